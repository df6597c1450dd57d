"""Input files for the forcing feed: a dependency-free NetCDF-4 / HDF5 reader and the raster input source.

The reference reads its inputs through NCDatasets / Rasters (`ext/TerrariumRastersExt/TerrariumRastersExt.jl`,
`examples/simulations/soil_heat_global.jl:30-37`).  None of the HDF5 / NetCDF libraries exist in this image, and the
files on this path are simple: numeric variables stored contiguously or in uncompressed / deflated chunks.  `Hdf5File`
parses exactly that subset of the HDF5 file format (superblock versions 0-3, version-1 and version-2 object headers,
symbol-table and compact-link groups, contiguous / compact / chunked (version-1 B-tree) layouts, the deflate, shuffle
and fletcher32 filters, fixed-point / floating-point / fixed-length string types, compact attributes) and refuses
anything else loudly.

`RasterInputSource` mirrors `RasterInputSource` of the reference's Rasters extension: a (possibly time-indexed) raster
on the full grid, gathered to the grid's columns through the mask's index map, handed to the library as a
device-resident time series (`trm_set_forcing_series`, TRM_TIME_RASTER) so that every step interpolates it on the
device: linear between the bracketing nodes, flat beyond the ends (TerrariumRastersExt.jl:96-121).
"""
import datetime
import re
import zlib

import numpy as np

UNDEF = 0xFFFFFFFFFFFFFFFF


class Hdf5FormatError(ValueError):
    pass


def lookup3(data: bytes, init: int = 0) -> int:
    """Bob Jenkins' lookup3 `hashlittle` -- the checksum of version-2 HDF5 metadata."""
    M = 0xFFFFFFFF
    rot = lambda x, k: ((x << k) & M) | (x >> (32 - k))
    length = len(data)
    a = b = c = (0xDEADBEEF + length + init) & M
    i = 0
    while length - i > 12:
        a = (a + int.from_bytes(data[i:i + 4], "little")) & M
        b = (b + int.from_bytes(data[i + 4:i + 8], "little")) & M
        c = (c + int.from_bytes(data[i + 8:i + 12], "little")) & M
        a = (a - c) & M; a ^= rot(c, 4); c = (c + b) & M
        b = (b - a) & M; b ^= rot(a, 6); a = (a + c) & M
        c = (c - b) & M; c ^= rot(b, 8); b = (b + a) & M
        a = (a - c) & M; a ^= rot(c, 16); c = (c + b) & M
        b = (b - a) & M; b ^= rot(a, 19); a = (a + c) & M
        c = (c - b) & M; c ^= rot(b, 4); b = (b + a) & M
        i += 12
    tail = data[i:] + b"\0" * 12
    if length - i > 0:
        a = (a + int.from_bytes(tail[0:4], "little")) & M
        b = (b + int.from_bytes(tail[4:8], "little")) & M
        c = (c + int.from_bytes(tail[8:12], "little")) & M
        c ^= b; c = (c - rot(b, 14)) & M
        a ^= c; a = (a - rot(c, 11)) & M
        b ^= a; b = (b - rot(a, 25)) & M
        c ^= b; c = (c - rot(b, 16)) & M
        a ^= c; a = (a - rot(c, 4)) & M
        b ^= a; b = (b - rot(a, 14)) & M
        c ^= b; c = (c - rot(b, 24)) & M
    return c


class Dataset:
    """One HDF5 dataset (a NetCDF variable): shape, numpy dtype, attributes; `read()` returns the array."""

    def __init__(self, f, name, messages):
        self.file, self.name = f, name
        self.attrs = {}
        self.shape, self.dtype, self.layout, self.filters = None, None, None, []
        for mtype, data in messages:
            if mtype == 0x01:
                self.shape = f._dataspace(data)
            elif mtype == 0x03:
                self.dtype = f._datatype(data)[0]
            elif mtype == 0x08:
                self.layout = f._layout(data)
            elif mtype == 0x0B:
                self.filters = f._filters(data)
            elif mtype == 0x0C:
                k, v = f._attribute(data)
                if k is not None:
                    self.attrs[k] = v
        if self.shape is None or self.dtype is None or self.layout is None:
            raise Hdf5FormatError(f"{name}: not a dataset (dataspace / datatype / layout message missing)")

    def _fill(self):
        """What storage that was never written reads as: the variable's `_FillValue` (NetCDF-4 stores it as the HDF5 fill
        value too), else 0."""
        fv = self.attrs.get("_FillValue")
        try:
            return np.asarray(fv, dtype=self.dtype).reshape(-1)[0] if fv is not None else self.dtype.type(0)
        except (TypeError, ValueError):
            return self.dtype.type(0)

    def read_cf(self) -> np.ndarray:
        """`read()` with the CF packing conventions applied, as Rasters.jl / NCDatasets.jl do by default on the reference side
        (examples/simulations/soil_heat_global_era5.jl reads ERA5 files through `Raster(...)`, `replace_missing(..., NaN)`):
        `_FillValue` / `missing_value` become NaN, then `value * scale_factor + add_offset` in float64.  Classic ERA5
        downloads are int16-packed: without this the raw integers would be taken for the physical values.  A variable without
        any of these attributes is returned as stored."""
        raw = self.read()
        a = self.attrs
        packed = any(k in a for k in ("scale_factor", "add_offset"))
        missing = [a[k] for k in ("_FillValue", "missing_value") if k in a and not isinstance(a[k], str)]
        if not packed and not missing:
            return raw
        out = raw.astype(np.float64)
        if missing:
            bad = np.zeros(raw.shape, dtype=bool)
            for m in missing:
                for v in np.asarray(m).reshape(-1):
                    bad |= (raw == v) if not (isinstance(v, np.floating) and np.isnan(v)) else np.isnan(out)
        if packed:
            out = out * float(np.asarray(a.get("scale_factor", 1.0)).reshape(-1)[0]) + float(np.asarray(a.get("add_offset", 0.0)).reshape(-1)[0])
        if missing:
            out[bad] = np.nan
        return out

    def read(self) -> np.ndarray:
        f, kind = self.file, self.layout[0]
        count = int(np.prod(self.shape, dtype=np.int64)) if self.shape else 1
        nbytes = count * self.dtype.itemsize
        if kind == "compact":
            raw = self.layout[1]
        elif kind == "contiguous":
            addr, size = self.layout[1], self.layout[2]
            if addr == UNDEF:
                return np.full(self.shape, self._fill(), self.dtype)      # never written: the fill value
            raw = f.buf[f.base + addr:f.base + addr + nbytes]
        else:
            return self._read_chunked()
        if len(raw) < nbytes:
            raise Hdf5FormatError(f"{self.name}: data truncated")
        return np.frombuffer(raw, dtype=self.dtype, count=count).reshape(self.shape).copy()

    def _unfilter(self, raw, mask):
        for n, (fid, values) in reversed(list(enumerate(self.filters))):
            if mask & (1 << n):
                continue
            if fid == 1:
                raw = zlib.decompress(raw)
            elif fid == 2:
                size = values[0] if values else self.dtype.itemsize
                raw = np.frombuffer(raw, np.uint8).reshape(size, -1).T.tobytes()
            elif fid == 3:
                raw = raw[:-4]
            else:
                raise Hdf5FormatError(f"{self.name}: filter {fid} is not supported (deflate, shuffle, fletcher32 are)")
        return raw

    def _read_chunked(self):
        f = self.file
        _, btree, chunk = self.layout
        rank = len(self.shape)
        out = np.full(self.shape, self._fill(), self.dtype)       # chunks that were never allocated read as the fill value
        if btree == UNDEF:
            return out
        cshape = chunk[:rank]
        for size, mask, offsets, addr in f._chunks(btree, rank):
            raw = self._unfilter(f.buf[f.base + addr:f.base + addr + size], mask)
            block = np.frombuffer(raw, dtype=self.dtype, count=int(np.prod(cshape))).reshape(cshape)
            sl_out = tuple(slice(o, min(o + c, s)) for o, c, s in zip(offsets, cshape, self.shape))
            sl_in = tuple(slice(0, s.stop - s.start) for s in sl_out)
            out[sl_out] = block[sl_in]
        return out


class Hdf5File:
    """Read-only view of an HDF5 / NetCDF-4 file held in memory.  `f[name]` -> Dataset, `name in f`, `f.keys()`."""

    SIG = b"\x89HDF\r\n\x1a\n"

    def __init__(self, path_or_bytes, verify_checksums=True):
        self.buf = path_or_bytes if isinstance(path_or_bytes, (bytes, bytearray)) else open(path_or_bytes, "rb").read()
        self.verify = verify_checksums
        start = 0
        while self.buf[start:start + 8] != self.SIG:      # the superblock sits at 0, 512, 1024, ...
            start = 512 if start == 0 else start * 2
            if start >= len(self.buf):
                raise Hdf5FormatError("not an HDF5 file (signature not found)")
        b, v = self.buf, self.buf[start + 8]
        if v in (0, 1):
            self.O, self.L = b[start + 13], b[start + 14]
            p = start + 24 + (4 if v == 1 else 0)
            self.base = self._uint(p, self.O)
            p += 4 * self.O
            root_header = self._uint(p + self.O, self.O)          # root symbol table entry: name offset, header address
        elif v in (2, 3):
            self.O, self.L = b[start + 9], b[start + 10]
            self.base = self._uint(start + 12, self.O)
            root_header = self._uint(start + 12 + 3 * self.O, self.O)
            end = start + 12 + 4 * self.O
            if self.verify and lookup3(b[start:end]) != self._uint(end, 4):
                raise Hdf5FormatError("superblock checksum mismatch")
        else:
            raise Hdf5FormatError(f"superblock version {v} is not supported")
        self.links = self._group_links(root_header)

    # ---- primitives -----------------------------------------------------------------------------------------------------
    def _uint(self, pos, n):
        return int.from_bytes(self.buf[pos:pos + n], "little")

    def keys(self):
        return list(self.links)

    def __contains__(self, name):
        return name in self.links

    def __getitem__(self, name) -> Dataset:
        if name not in self.links:
            raise KeyError(f"{name!r}; the file holds {sorted(self.links)}")
        return Dataset(self, name, self._object_header(self.links[name]))

    # ---- object headers ---------------------------------------------------------------------------------------------------
    def _object_header(self, addr):
        """[(message type, message bytes)] of the object header at `addr`, continuation blocks included."""
        b, p = self.buf, self.base + addr
        msgs = []
        if b[p:p + 4] == b"OHDR":
            flags = b[p + 5]
            q = p + 6 + (16 if flags & 0x20 else 0) + (4 if flags & 0x10 else 0)
            nsz = 1 << (flags & 3)
            size0 = self._uint(q, nsz)
            q += nsz
            if self.verify and lookup3(b[p:q + size0]) != self._uint(q + size0, 4):
                raise Hdf5FormatError(f"object header at {addr}: checksum mismatch")
            blocks = [(q, q + size0)]
            while blocks:
                q, end = blocks.pop(0)
                while q + 4 <= end:
                    mtype, msize, mflags = b[q], self._uint(q + 1, 2), b[q + 3]
                    q += 4 + (2 if flags & 0x04 else 0)
                    data = b[q:q + msize]
                    q += msize
                    if mtype == 0x10:
                        off, length = self._uint(q - msize, self.O), self._uint(q - msize + self.O, self.L)
                        cp = self.base + off
                        if b[cp:cp + 4] != b"OCHK":
                            raise Hdf5FormatError("continuation block without OCHK signature")
                        if self.verify and lookup3(b[cp:cp + length - 4]) != self._uint(cp + length - 4, 4):
                            raise Hdf5FormatError("continuation block checksum mismatch")
                        blocks.append((cp + 4, cp + length - 4))
                    elif mtype != 0:
                        msgs.append((mtype, data))
            return msgs
        if b[p] != 1:
            raise Hdf5FormatError(f"object header at {addr}: unknown version {b[p]}")
        nmsg, hsize = self._uint(p + 2, 2), self._uint(p + 8, 4)
        blocks = [(p + 16, p + 16 + hsize)]
        while blocks and len(msgs) < 4 * nmsg + 64:
            q, end = blocks.pop(0)
            while q + 8 <= end:
                mtype, msize = self._uint(q, 2), self._uint(q + 2, 2)
                data = b[q + 8:q + 8 + msize]
                q += 8 + msize
                if mtype == 0x10:
                    off, length = int.from_bytes(data[:self.O], "little"), int.from_bytes(data[self.O:self.O + self.L], "little")
                    blocks.append((self.base + off, self.base + off + length))
                elif mtype != 0:
                    msgs.append((mtype, data))
        return msgs

    # ---- groups -----------------------------------------------------------------------------------------------------------
    def _group_links(self, addr):
        links = {}
        for mtype, data in self._object_header(addr):
            if mtype == 0x06:                       # link message (new-style group, compact storage)
                name, target = self._link(data)
                if target is not None:
                    links[name] = target
            elif mtype == 0x02:                     # link info: dense storage lives in a fractal heap
                flags = data[1]
                heap = int.from_bytes(data[2 + (8 if flags & 1 else 0):][:self.O], "little")
                if heap != UNDEF:
                    raise Hdf5FormatError("groups with densely stored links (fractal heap) are not supported")
            elif mtype == 0x11:                     # symbol table (old-style group)
                btree = int.from_bytes(data[:self.O], "little")
                heap = int.from_bytes(data[self.O:2 * self.O], "little")
                links.update(self._symbol_table(btree, heap))
        return links

    def _link(self, data):
        flags = data[1]
        p = 2
        ltype = 0
        if flags & 0x08:
            ltype = data[p]; p += 1
        if flags & 0x04:
            p += 8
        if flags & 0x10:
            p += 1
        nsz = 1 << (flags & 3)
        n = int.from_bytes(data[p:p + nsz], "little"); p += nsz
        name = bytes(data[p:p + n]).decode("utf-8"); p += n
        if ltype != 0:
            return name, None                       # soft / external links are not followed
        return name, int.from_bytes(data[p:p + self.O], "little")

    def _symbol_table(self, btree, heap):
        b = self.buf
        hp = self.base + heap
        if b[hp:hp + 4] != b"HEAP":
            raise Hdf5FormatError("local heap signature missing")
        seg = self.base + self._uint(hp + 8 + 2 * self.L, self.O)
        out = {}

        def node(addr):
            p = self.base + addr
            if b[p:p + 4] == b"SNOD":
                n = self._uint(p + 6, 2)
                q = p + 8
                for _ in range(n):
                    off, hdr = self._uint(q, self.O), self._uint(q + self.O, self.O)
                    end = b.index(b"\0", seg + off)
                    out[bytes(b[seg + off:end]).decode("utf-8")] = hdr
                    q += 2 * self.O + 24
                return
            if b[p:p + 4] != b"TREE":
                raise Hdf5FormatError("group B-tree signature missing")
            used = self._uint(p + 6, 2)
            q = p + 8 + 2 * self.O + self.L
            for _ in range(used):
                node(self._uint(q, self.O))
                q += self.O + self.L
        node(btree)
        return out

    # ---- messages ---------------------------------------------------------------------------------------------------------
    def _dataspace(self, d):
        v, rank, flags = d[0], d[1], d[2]
        p = 8 if v == 1 else 4
        return tuple(int.from_bytes(d[p + i * self.L:p + (i + 1) * self.L], "little") for i in range(rank))

    def _datatype(self, d):
        """(numpy dtype, bytes consumed)"""
        cls, bits0, size = d[0] & 0x0F, d[4 - 3], int.from_bytes(d[4:8], "little")
        order = ">" if bits0 & 1 else "<"
        if cls == 0:
            return np.dtype(f"{order}{'i' if bits0 & 0x08 else 'u'}{size}"), 12
        if cls == 1:
            return np.dtype(f"{order}f{size}"), 20
        if cls == 3:
            return np.dtype(f"S{size}"), 8
        raise Hdf5FormatError(f"datatype class {cls} is not supported (fixed-point, floating-point, fixed strings are)")

    def _layout(self, d):
        v, cls = d[0], d[1]
        if v == 3:
            if cls == 0:
                n = int.from_bytes(d[2:4], "little")
                return ("compact", bytes(d[4:4 + n]))
            if cls == 1:
                return ("contiguous", int.from_bytes(d[2:2 + self.O], "little"), int.from_bytes(d[2 + self.O:2 + self.O + self.L], "little"))
            if cls == 2:
                nd = d[2]
                btree = int.from_bytes(d[3:3 + self.O], "little")
                dims = [int.from_bytes(d[3 + self.O + 4 * i:7 + self.O + 4 * i], "little") for i in range(nd)]
                return ("chunked", btree, dims)
        if v == 4 and cls == 1:
            return ("contiguous", int.from_bytes(d[2:2 + self.O], "little"), int.from_bytes(d[2 + self.O:2 + self.O + self.L], "little"))
        if v == 4 and cls == 0:
            n = int.from_bytes(d[2:4], "little")
            return ("compact", bytes(d[4:4 + n]))
        raise Hdf5FormatError(f"data layout version {v} class {cls} is not supported")

    def _filters(self, d):
        v, n = d[0], d[1]
        p = 8 if v == 1 else 2
        out = []
        for _ in range(n):
            fid = int.from_bytes(d[p:p + 2], "little"); p += 2
            nlen = 0
            if v == 1 or fid >= 256:
                nlen = int.from_bytes(d[p:p + 2], "little"); p += 2
            p += 2
            ncv = int.from_bytes(d[p:p + 2], "little"); p += 2
            p += (nlen + 7) // 8 * 8 if v == 1 else nlen
            vals = [int.from_bytes(d[p + 4 * i:p + 4 * i + 4], "little") for i in range(ncv)]
            p += 4 * ncv + (4 if v == 1 and ncv % 2 else 0)
            out.append((fid, vals))
        return out

    def _attribute(self, d):
        v = d[0]
        nsz, tsz, ssz = (int.from_bytes(d[2 + 2 * i:4 + 2 * i], "little") for i in range(3))
        p = 8 if v < 3 else 9
        pad = (lambda n: (n + 7) // 8 * 8) if v == 1 else (lambda n: n)
        name = bytes(d[p:p + nsz]).split(b"\0")[0].decode("utf-8", "replace"); p += pad(nsz)
        try:
            dtype, _ = self._datatype(d[p:p + tsz])
        except Hdf5FormatError:
            return None, None
        p += pad(tsz)
        shape = self._dataspace(d[p:p + ssz]) if ssz >= 4 and d[p + 1] else ()
        p += pad(ssz)
        count = int(np.prod(shape)) if shape else 1
        raw = d[p:p + count * dtype.itemsize]
        if len(raw) < count * dtype.itemsize:
            return None, None
        arr = np.frombuffer(bytes(raw), dtype=dtype, count=count)
        if dtype.kind == "S":
            return name, arr[0].split(b"\0")[0].decode("utf-8", "replace")
        return name, (arr.reshape(shape) if shape else arr[0])

    def _chunks(self, addr, rank):
        """(size, filter mask, offsets[rank], address) of every chunk under the version-1 B-tree node at `addr`."""
        b, p = self.buf, self.base + addr
        if b[p:p + 4] != b"TREE" or b[p + 4] != 1:
            raise Hdf5FormatError("chunk index is not a version-1 B-tree")
        level, used = b[p + 5], self._uint(p + 6, 2)
        q = p + 8 + 2 * self.O
        keysz = 8 + 8 * (rank + 1)
        for _ in range(used):
            size, mask = self._uint(q, 4), self._uint(q + 4, 4)
            offsets = [self._uint(q + 8 + 8 * i, 8) for i in range(rank)]
            child = self._uint(q + keysz, self.O)
            q += keysz + self.O
            if level == 0:
                yield size, mask, offsets, child
            else:
                yield from self._chunks(child, rank)


# ---- CF time axes ----------------------------------------------------------------------------------------------------------
_UNIT_SECONDS = dict(second=1.0, seconds=1.0, sec=1.0, secs=1.0, s=1.0, minute=60.0, minutes=60.0, min=60.0, hour=3600.0,
                     hours=3600.0, hr=3600.0, h=3600.0, day=86400.0, days=86400.0, d=86400.0)


def decode_time_axis(values, units=None):
    """Time coordinate -> (seconds since the axis' epoch as float64, epoch as datetime | None).
    `units` is a CF string "<unit> since YYYY-MM-DD[ hh:mm:ss]"; without one the values are taken as seconds."""
    t = np.asarray(values, dtype=np.float64)
    if not units:
        return t, None
    m = re.match(r"\s*(\w+)\s+since\s+(\d{4})-(\d{1,2})-(\d{1,2})(?:[T ](\d{1,2}):(\d{1,2})(?::(\d{1,2})(?:\.\d*)?)?)?", units)
    if not m or m.group(1).lower() not in _UNIT_SECONDS:
        raise ValueError(f"cannot interpret the time units {units!r}")
    y, mo, d, hh, mm, ss = (int(g) if g else 0 for g in m.groups()[1:])
    return t * _UNIT_SECONDS[m.group(1).lower()], datetime.datetime(y, mo, d, hh, mm, ss)


# ---- the raster input source -------------------------------------------------------------------------------------------------
class RasterInputSource:
    """`InputSource(grid::ColumnRingGrid, raster; name, reftime)` of ext/TerrariumRastersExt (lines 21-52).

    `data`: the raster on the FULL grid, `mask.shape` per time level -- `[ny][nx]` (static) or `[nt][ny][nx]` with
    `times[nt]` in seconds relative to `reftime` (default: the first time, TerrariumRastersExt.jl:132-137).
    `columns()` gathers the masked points in ring order (`view(raster, idxmap)`, line 47): `[Nh]` or `[nt][Nh]`."""

    def __init__(self, grid, data, name, times=None, reftime=None):
        data = np.asarray(data)
        self.grid, self.name = grid, name
        mask = np.asarray(grid.mask, dtype=bool)
        if data.shape[-mask.ndim:] != mask.shape:
            raise ValueError(f"raster {name!r} has shape {data.shape}; its trailing axes must match the grid mask {mask.shape}")
        self.static = times is None
        if self.static:
            if data.ndim != mask.ndim:
                raise ValueError("a raster without a time axis must have exactly the mask's dimensions")
            self.times = None
        else:
            t = np.asarray(times, dtype=np.float64)
            if data.ndim != mask.ndim + 1 or data.shape[0] != t.size:
                raise ValueError("a time-indexed raster is [nt] + mask.shape with one time per level")
            self.times = t - (t[0] if reftime is None else float(reftime))
        self.data = data

    @classmethod
    def from_netcdf(cls, grid, path, variable, time_variable="time", reftime=None, name=None):
        """Reads `variable` (dimensions [time,] y, x in file order) and its time coordinate from a NetCDF-4 file."""
        f = Hdf5File(path)
        ds = f[variable]
        data = ds.read_cf()      # scale_factor / add_offset / _FillValue applied, as Rasters does by default
        mask_ndim = np.asarray(grid.mask).ndim
        times = None
        if data.ndim == mask_ndim + 1:
            tv = f[time_variable]
            times, _ = decode_time_axis(tv.read(), tv.attrs.get("units"))
        return cls(grid, data, name or variable, times=times, reftime=reftime)

    def columns(self) -> np.ndarray:
        return self.grid.gather(self.data)

    def attach(self, state):
        """Hands the source to a DeviceState: a static raster is set once (initialize_from_raster!, line 67), a
        time-indexed one becomes a device-resident series evaluated by every step (update_from_raster!, lines 96-121)."""
        cols = self.columns()
        if self.static:
            state.set_forcing(self.name, cols)
        else:
            state.set_forcing_series(self.name, self.times, cols, "raster")

    def attach_boundary(self, state, var, side, kind):
        """The source feeds a boundary condition -- `PrescribedSurfaceTemperature(:Tair)` with `InputSource(grid, raster; name
        = :Tair)` (examples/simulations/soil_heat_global_era5.jl:31-44): same update rule, evaluated into the boundary values."""
        cols = self.columns()
        if self.static:
            state.set_bc(var, side, kind, cols)
        else:
            state.set_bc_series(var, side, kind, self.times, cols, "raster")
