#!/usr/bin/env python3
"""Soil heat conduction at global scale on one MI355X -- the host-side mirror of the reference's
examples/simulations/soil_heat_global.jl, line for line where the interface allows:

    land mask (full Gaussian grid N72, > 50 % land)  -> ColumnRingGrid(ExponentialSpacing(N = 30), mask)
    SoilModel(grid), ForwardEuler
    initial temperature  T0(lat) - 0.05 z,  T0(lat) = 20 - |40 sin(lat)|
    surface boundary     T0 + 10 sin(2 pi t / day - lon)            (PrescribedSurfaceTemperature)
    timestep!, then run!(period = 12 h, dt = 600 s); surface layer scattered back to the ring grid

The one difference: the periodic boundary function is sampled into a FieldTimeSeries (here every 600 s, i.e. on
every step time, so the sampled values ARE the function's values) which lives on the device; `run` is then a single
library call instead of one host callback per step.

    python examples/soil_heat_global.py [--mask N72|N145] [--hours 12] [--float32]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import terrarium_jl_amd as trm  # noqa: E402

DAY = 86400.0


def mean_annual_temperature(lat):
    return 20.0 - np.abs(40.0 * np.sin(lat))     # maximum at the equator


def build(mask_name="N72", dtype=np.float64, hours=12.0, dt=600.0):
    land_mask = trm.masks.load_land_mask(mask_name)                       # land_sea_frac .> 0.5
    grid = trm.ColumnRingGrid(trm.ExponentialSpacing(N=30), land_mask, dtype=dtype)
    lat_masked, lon_masked = trm.masks.masked_latlon(land_mask)
    model = trm.SoilModel(grid)
    T0 = mean_annual_temperature(lat_masked)

    def initial_soil_temperature(x, z):
        return T0[int(round(x)) - 1] - 0.05 * z

    def periodic_bc(t, amplitude=10.0):
        return T0 + amplitude * np.sin(2.0 * np.pi * t / DAY - lon_masked)

    nsteps = int(hours * 3600.0 // dt)
    times = dt * np.arange(nsteps + 2)
    bc = trm.PrescribedSurfaceTemperature("T_ub", trm.FieldTimeSeries.from_function(periodic_bc, times))
    integrator = trm.initialize(model, trm.ForwardEuler(), boundary_conditions=trm.merge_boundary_conditions(bc),
                                initializers=dict(temperature=initial_soil_temperature))
    return grid, integrator, nsteps, periodic_bc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mask", default="N72", choices=["N72", "N145"])
    ap.add_argument("--hours", type=float, default=12.0)
    ap.add_argument("--float32", action="store_true")
    args = ap.parse_args()
    dt = 600.0
    grid, integrator, nsteps, _ = build(args.mask, np.float32 if args.float32 else np.float64, args.hours, dt)
    T_surface_initial = grid.scatter(integrator.state.get("temperature")[-1])
    trm.timestep(integrator, dt)                     # quick check: one step
    t0 = time.perf_counter()
    trm.run(integrator, steps=nsteps - 1, dt=dt)     # ... then the rest of the period in one call
    wall = time.perf_counter() - t0
    T_surface = grid.scatter(integrator.state.get("temperature")[-1])
    land = grid.mask
    print(f"{grid.num_columns} land columns x 30 levels, {nsteps} steps of {dt:.0f} s: {wall * 1e3:.1f} ms "
          f"({grid.num_columns * (nsteps - 1) / wall / 1e9:.2f} G column-steps/s incl. host)")
    print(f"uppermost layer: initial {np.nanmin(T_surface_initial):.2f} .. {np.nanmax(T_surface_initial):.2f} degC, "
          f"after {args.hours:g} h {T_surface[land].min():.2f} .. {T_surface[land].max():.2f} degC; "
          f"status flags {integrator.state.status()}")
    return T_surface


if __name__ == "__main__":
    main()
