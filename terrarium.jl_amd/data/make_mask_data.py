"""Regenerates the packaged land masks from the reference's input files (needs /root/reference; run in the build
container):  python terrarium.jl_amd/data/make_mask_data.py

The reference ships `inputs/era5-land_land_sea_mask_N{72,145}.nc` (NetCDF-4: `lsm(time, lat, lon)`, float64) and builds
its mask as `lsm .> 0.5` (examples/simulations/soil_heat_global.jl:30-37).  The package keeps the thresholded mask only,
packed to bits (5 KB / 21 KB): data, not code.  The files are read with the package's own reader (terrarium.jl_amd/io.py)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import terrarium_jl_amd as trm  # noqa: E402

REF_INPUTS = "/root/reference/inputs"
EXPECTED_LAND = {"N72": 14017, "N145": 56951}


def main():
    for name, expected in EXPECTED_LAND.items():
        mask = trm.masks.land_mask_from_netcdf(os.path.join(REF_INPUTS, f"era5-land_land_sea_mask_{name}.nc"))
        assert int(mask.sum()) == expected, (name, int(mask.sum()))
        np.savez_compressed(os.path.join(HERE, f"era5_land_mask_{name}.npz"), packed=np.packbits(mask.ravel()),
                            shape=np.array(mask.shape, dtype=np.int32), land_count=np.int64(mask.sum()))
        print(name, mask.shape, "land columns", int(mask.sum()))


if __name__ == "__main__":
    sys.exit(main())
