"""Generate the packed land-mask fixtures from the reference's input data files.

Run once in the build container (needs /root/reference); the outputs
tests/golden/era5_land_mask_N{72,145}.npz are committed.  The .nc files are
NetCDF-4/HDF5 holding `lsm(time, lat, lon)` as ONE uncompressed contiguous
chunk of little-endian float64, so no HDF5 library is
needed: locate the chunk through the v1 B-tree node ("TREE") and frombuffer it.
The fixture is data only: `lsm > 0.5` packed to bits plus the shape, using the
threshold of examples/simulations/soil_heat_global.jl:37.
"""
import os
import struct
import sys

import numpy as np

REF_INPUTS = "/root/reference/inputs"
SHAPES = {"N72": (144, 288), "N145": (290, 580)}
EXPECTED_LAND = {"N72": 14017, "N145": 56951}


def find_chunk(buf: bytes, nbytes: int):
    """Scan v1 B-tree chunk nodes for the entry whose chunk size is `nbytes`."""
    pos = 0
    while True:
        pos = buf.find(b"TREE", pos)
        if pos < 0:
            raise RuntimeError("no chunk B-tree entry of the expected size")
        node_type, level, entries = struct.unpack_from("<BBH", buf, pos + 4)
        if node_type == 1 and level == 0:
            # header: sig(4) type(1) level(1) entries(2) left(8) right(8); then keys/children
            off = pos + 24
            for _ in range(entries):
                size, _filter_mask = struct.unpack_from("<II", buf, off)
                # key = size(4) mask(4) offsets[(rank+1)*8]; rank = 3 -> 4 offsets
                addr, = struct.unpack_from("<Q", buf, off + 8 + 4 * 8)
                if size == nbytes:
                    return addr
                off += 8 + 4 * 8 + 8
        pos += 4


def main():
    out_dir = os.path.dirname(os.path.abspath(__file__))
    for name, (nlat, nlon) in SHAPES.items():
        path = os.path.join(REF_INPUTS, f"era5-land_land_sea_mask_{name}.nc")
        buf = open(path, "rb").read()
        assert buf[:4] == b"\x89HDF"
        nbytes = nlat * nlon * 8
        addr = find_chunk(buf, nbytes)
        lsm = np.frombuffer(buf, dtype="<f8", count=nlat * nlon, offset=addr).reshape(nlat, nlon)
        mask = lsm > 0.5
        assert int(mask.sum()) == EXPECTED_LAND[name], (name, int(mask.sum()))
        np.savez_compressed(os.path.join(out_dir, f"era5_land_mask_{name}.npz"),
                            packed=np.packbits(mask.ravel()), shape=np.array([nlat, nlon], dtype=np.int32),
                            land_count=np.int64(mask.sum()))
        print(name, "chunk offset", addr, "land columns", int(mask.sum()))


if __name__ == "__main__":
    sys.exit(main())
