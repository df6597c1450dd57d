#!/usr/bin/env python3
"""Global soil heat conduction driven by a gridded 2 m air temperature record -- the host-side mirror of the
reference's examples/simulations/soil_heat_global_era5.jl:

    land mask (N72 Gaussian grid, > 50 % land)        -> ColumnRingGrid(ExponentialSpacing(N = 30), mask)
    Raster("...2m_temperature...nc")                  -> RasterInputSource.from_netcdf(grid, path, variable)   [K -> degC]
    PrescribedSurfaceTemperature(:Tair)               -> the raster source feeds the surface boundary value, interpolated
                                                         in time on the device (x1 + eps (x2 - x1) / dt, flat ends)
    temperature = Tsurf_0 - 0.02 z, saturation = 1;   timestep!, run!(period = 10 days, dt = 120 s)

The NetCDF-4 file (dimensions time, lat, lon matching the mask; CF time units) is read by the library's own HDF5
reader -- contiguous, compact or chunked (deflate / shuffle / fletcher32) layouts.

    python examples/soil_heat_global_era5.py --temperature-file era5_land_2m_temperature_2023_N72.nc [--variable t2m] [--days 10]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import terrarium_jl_amd as trm  # noqa: E402

DAY = 86400.0


def build(path, variable="t2m", mask_name="N72", dtype=np.float32, kelvin=True, time_variable="time"):
    land_mask = trm.masks.load_land_mask(mask_name)
    grid = trm.ColumnRingGrid(trm.ExponentialSpacing(N=30), land_mask, dtype=dtype)
    Tair = trm.RasterInputSource.from_netcdf(grid, path, variable, time_variable=time_variable, name="Tair")
    if kelvin:
        Tair.data = Tair.data.astype(np.float64) - 273.15
    Tsurf_0 = Tair.columns()[0] if not Tair.static else Tair.columns()
    bcs = trm.merge_boundary_conditions(trm.PrescribedSurfaceTemperature("Tair", Tair))
    initializers = dict(temperature=lambda x, z: Tsurf_0[int(round(x)) - 1] - 0.02 * z, saturation_water_ice=1.0)
    integrator = trm.initialize(trm.SoilModel(grid), trm.ForwardEuler(), boundary_conditions=bcs, initializers=initializers)
    return grid, integrator, Tair


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--temperature-file", required=True)
    ap.add_argument("--variable", default="t2m")
    ap.add_argument("--mask", default="N72", choices=["N72", "N145"])
    ap.add_argument("--days", type=float, default=10.0)
    ap.add_argument("--celsius", action="store_true", help="the file already holds degC")
    args = ap.parse_args()
    grid, integrator, _ = build(args.temperature_file, args.variable, args.mask, kelvin=not args.celsius)
    trm.timestep(integrator)
    integrator.state.set_option("steps_per_launch", 50)      # the series is interpolated inside the resident-column program
    t0 = time.perf_counter()
    trm.run(integrator, period=args.days * DAY, dt=120.0)
    wall = time.perf_counter() - t0
    T_surface = grid.scatter(integrator.state.temperature[-1])
    print(f"{grid.num_columns} columns, {args.days} days at dt = 120 s in {wall:.2f} s; "
          f"surface-layer temperature {np.nanmin(T_surface):.2f} .. {np.nanmax(T_surface):.2f} degC; status {integrator.state.status()}")


if __name__ == "__main__":
    main()
