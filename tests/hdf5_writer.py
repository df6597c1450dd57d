"""A tiny HDF5 (NetCDF-4 style) WRITER for the tests of terrarium.jl_amd/io.py: builds files byte by byte, following the
HDF5 file-format specification -- superblock version 2, version-2 object headers with link messages, datasets with
contiguous, compact or chunked (version-1 B-tree) layouts, optional shuffle + deflate filters, fixed-string attributes.
Test infrastructure only."""
import struct
import zlib

import numpy as np

from terrarium_jl_amd.io import lookup3

UNDEF = 0xFFFFFFFFFFFFFFFF


def _msg(mtype, body, flags=0):
    return struct.pack("<BHB", mtype, len(body), flags) + body


def _ohdr(messages):
    body = b"".join(messages)
    head = b"OHDR" + bytes([2, 0x02]) + struct.pack("<I", len(body))    # flags: 4-byte chunk-0 size, no times
    blob = head + body
    return blob + struct.pack("<I", lookup3(blob))


def _dataspace(shape):
    return bytes([2, len(shape), 0, 1]) + b"".join(struct.pack("<Q", n) for n in shape)


def _datatype(dtype):
    dtype = np.dtype(dtype)
    if dtype.kind == "f":
        if dtype.itemsize == 4:
            props = struct.pack("<HHBBBBI", 0, 32, 23, 8, 0, 23, 127)
            bits = bytes([0x20, 31, 0])
        else:
            props = struct.pack("<HHBBBBI", 0, 64, 52, 11, 0, 52, 1023)
            bits = bytes([0x20, 63, 0])
        return bytes([0x11]) + bits + struct.pack("<I", dtype.itemsize) + props
    if dtype.kind in "iu":
        bits = bytes([0x08 if dtype.kind == "i" else 0x00, 0, 0])
        return bytes([0x10]) + bits + struct.pack("<I", dtype.itemsize) + struct.pack("<HH", 0, 8 * dtype.itemsize)
    if dtype.kind == "S":
        return bytes([0x13, 0, 0, 0]) + struct.pack("<I", dtype.itemsize)
    raise TypeError(dtype)


def _attribute(name, text):
    nm = name.encode() + b"\0"
    if not isinstance(text, str):                                   # a numeric attribute (scale_factor, _FillValue, ...): scalar
        arr = np.asarray(text)
        val, dt, ds = arr.tobytes(), _datatype(arr.dtype), bytes([2, 0, 0, 0])
        return bytes([3, 0]) + struct.pack("<HHH", len(nm), len(dt), len(ds)) + bytes([0]) + nm + dt + ds + val
    val = text.encode() + b"\0"
    dt, ds = _datatype(f"S{len(val)}"), bytes([2, 0, 0, 0])        # scalar dataspace
    return bytes([3, 0]) + struct.pack("<HHH", len(nm), len(dt), len(ds)) + bytes([0]) + nm + dt + ds + val


class Writer:
    def __init__(self):
        self.blob = bytearray(48)        # the superblock goes here at the end
        self.links = []

    def _put(self, data, align=8):
        while len(self.blob) % align:
            self.blob.append(0)
        addr = len(self.blob)
        self.blob += data
        return addr

    def dataset(self, name, array, layout="contiguous", chunks=None, deflate=False, shuffle=False, attrs=None):
        a = np.ascontiguousarray(array)
        msgs = [_msg(0x01, _dataspace(a.shape)), _msg(0x03, _datatype(a.dtype), 1)]
        msgs.append(_msg(0x05, bytes([3, 0x09])))                      # fill value: version 3, allocate early, undefined
        if layout == "compact":
            raw = a.tobytes()
            msgs.append(_msg(0x08, bytes([3, 0]) + struct.pack("<H", len(raw)) + raw))
        elif layout == "contiguous":
            addr = self._put(a.tobytes())
            msgs.append(_msg(0x08, bytes([3, 1]) + struct.pack("<QQ", addr, a.nbytes)))
        else:
            rank = a.ndim
            filters = []
            if shuffle:
                filters.append(struct.pack("<HHH", 2, 0, 1) + struct.pack("<I", a.dtype.itemsize))
            if deflate:
                filters.append(struct.pack("<HHH", 1, 0, 1) + struct.pack("<I", 6))
            if filters:
                msgs.append(_msg(0x0B, bytes([2, len(filters)]) + b"".join(filters)))
            entries = []
            grid = [range(0, s, c) for s, c in zip(a.shape, chunks)]
            for idx in np.ndindex(*[len(g) for g in grid]):
                off = [g[i] for g, i in zip(grid, idx)]
                block = np.zeros(chunks, a.dtype)
                sl = tuple(slice(o, min(o + c, s)) for o, c, s in zip(off, chunks, a.shape))
                block[tuple(slice(0, s.stop - s.start) for s in sl)] = a[sl]
                raw = block.tobytes()
                if shuffle:
                    raw = np.frombuffer(raw, np.uint8).reshape(-1, a.dtype.itemsize).T.tobytes()
                if deflate:
                    raw = zlib.compress(raw, 6)
                entries.append((len(raw), off, self._put(raw)))
            key = lambda size, off: struct.pack("<II", size, 0) + b"".join(struct.pack("<Q", o) for o in off) + struct.pack("<Q", 0)
            node = b"TREE" + bytes([1, 0]) + struct.pack("<H", len(entries)) + struct.pack("<QQ", UNDEF, UNDEF)
            for size, off, addr in entries:
                node += key(size, off) + struct.pack("<Q", addr)
            node += key(0, list(a.shape))                               # the final key
            btree = self._put(node)
            dims = b"".join(struct.pack("<I", c) for c in list(chunks) + [a.dtype.itemsize])
            msgs.append(_msg(0x08, bytes([3, 2, rank + 1]) + struct.pack("<Q", btree) + dims))
        for k, v in (attrs or {}).items():
            msgs.append(_msg(0x0C, _attribute(k, v)))
        self.links.append((name, self._put(_ohdr(msgs))))

    def tobytes(self):
        links = []
        for name, addr in self.links:
            nm = name.encode()
            links.append(_msg(0x06, bytes([1, 0x00, len(nm)]) + nm + struct.pack("<Q", addr)))
        link_info = _msg(0x02, bytes([0, 0]) + struct.pack("<QQ", UNDEF, UNDEF))
        root = self._put(_ohdr([link_info] + links))
        sb = b"\x89HDF\r\n\x1a\n" + bytes([2, 8, 8, 0]) + struct.pack("<QQQQ", 0, UNDEF, len(self.blob), root)
        sb += struct.pack("<I", lookup3(sb))
        self.blob[:len(sb)] = sb
        return bytes(self.blob)
