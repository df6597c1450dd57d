#!/usr/bin/env python3
"""Soil heat conduction in a 1-D column -- the host-side mirror of the reference's examples/simulations/soil_heat_column.jl:

    ColumnGrid(Float32, ExponentialSpacing(N = 10))
    SoilInitializer(energy = QuasiThermalSteadyState(T0 = -1), hydrology = ConstantSaturation(sat = 1))
    SoilModel(grid; initializer), PrescribedSurfaceTemperature(:T_ub, 1.0), ForwardEuler
    timestep!, run!(period = 3 days); temperature and liquid fraction profiles

    python examples/soil_heat_column.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import terrarium_jl_amd as trm  # noqa: E402

DAY = 86400.0


def build(dtype=np.float32):
    grid = trm.ColumnGrid(trm.ExponentialSpacing(N=10), dtype=dtype)
    initializer = trm.SoilInitializer(energy=trm.QuasiThermalSteadyState(T0=-1.0), hydrology=trm.ConstantSaturation(sat=1.0))
    model = trm.SoilModel(grid, initializer=initializer)
    boundary_conditions = trm.merge_boundary_conditions(trm.PrescribedSurfaceTemperature("T_ub", 1.0))
    return trm.initialize(model, trm.ForwardEuler(), boundary_conditions=boundary_conditions)


def main():
    integrator = build()
    trm.timestep(integrator)
    trm.run(integrator, period=3 * DAY)
    T = integrator.state.temperature[:, 0]
    f = integrator.state.liquid_water_fraction[:, 0]
    zs = integrator.state.grid.z_centers()
    print("  depth / m   temperature / degC   liquid fraction")
    for z, t, l in zip(zs[::-1], T[::-1], f[::-1]):
        print(f"  {z:9.3f}   {t:18.4f}   {l:15.4f}")
    print(f"model time {trm.current_time(integrator) / DAY:.3f} days, status {integrator.state.status()}")


if __name__ == "__main__":
    main()
