#!/usr/bin/env python3
"""Coupling the soil model to an atmosphere that lives on the same GPU -- the structure of the reference's
examples/simulations/speedy_dry_land.jl (`TerrariumDryLand <: Speedy.AbstractDryLand`) with the SpeedyWeather model
replaced by a toy energy-balance atmosphere written in PyTorch (SpeedyWeather is a Julia package):

    ColumnRingGrid(Float32, ExponentialSpacing(N = 30, dz_min = 0.05), FullGaussianGrid(24))     -- every point is land
    SoilModel + PrescribedSurfaceTemperature(:air_temperature), the input written by the atmosphere each coupling step
    per coupling step:   atmosphere -> air temperature (K) -> boundary values (degC)   [zero-copy, device to device]
                         run!(integrator, period = dt_atmosphere, dt = 300 s)
                         surface soil temperature -> atmosphere                          [zero-copy]

The exchange never leaves the device: `state.bc_device_array("temperature", "top")` and `state.device_array("temperature")`
are views of the library's buffers (`__cuda_array_interface__`), written and read by torch kernels on the same stream.

    python examples/coupled_dry_land.py [--days 30]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import terrarium_jl_amd as trm  # noqa: E402

DAY = 86400.0


class TerrariumDryLand:
    """What speedy_dry_land.jl:19-66 implements: initialize (soil temperature -> atmosphere) and timestep."""

    def __init__(self, integrator):
        import torch
        self.torch = torch
        self.integrator = integrator
        st = integrator.state
        self.bc = torch.as_tensor(st.bc_device_array("temperature", "top"), device="cuda")          # [Nh], degC
        T = torch.as_tensor(st.device_array("temperature"), device="cuda")                           # [Nh][pitch], z-fastest
        self.T_surface = T[:, st.grid.Nz - 1]                                                        # view of the top cells

    def initialize(self):
        return self.T_surface + 273.15

    def timestep(self, air_temperature_K, dt_atmosphere, dt_land=300.0):
        self.bc.copy_(air_temperature_K - 273.15)          # set!(state.inputs.air_temperature, Tair - 273.15)
        self.torch.cuda.current_stream().synchronize()     # the write is ordered before the library's stream
        trm.run(self.integrator, period=dt_atmosphere, dt=dt_land)
        return self.T_surface + 273.15                     # progn.land.soil_temperature


def build(nlat_half=24, dtype=np.float32):
    nlat, nlon = 2 * nlat_half, 4 * nlat_half
    mask = np.ones((nlat, nlon), dtype=bool)               # RockyPlanetMask: land everywhere
    grid = trm.ColumnRingGrid(trm.ExponentialSpacing(N=30, dz_min=0.05), mask, dtype=dtype)
    model = trm.SoilModel(grid, initializer=trm.SoilInitializer())
    bcs = trm.merge_boundary_conditions(trm.PrescribedSurfaceTemperature("air_temperature", 0.0))
    integrator = trm.initialize(model, trm.ForwardEuler(), boundary_conditions=bcs)
    lat = np.repeat(np.linspace(np.pi / 2, -np.pi / 2, nlat + 2)[1:-1], nlon)
    lon = np.tile(np.linspace(0.0, 2 * np.pi, nlon, endpoint=False), nlat)
    return grid, integrator, lat, lon


def toy_atmosphere(torch, lat, lon, t, T_soil_K, T_air_K, dt):
    """Relaxation towards a radiative equilibrium with a diurnal cycle, plus exchange with the soil surface."""
    T_eq = 288.0 - 40.0 * torch.sin(lat) ** 2 + 8.0 * torch.cos(lat) * torch.sin(2 * np.pi * t / DAY - lon)
    return T_air_K + dt * ((T_eq - T_air_K) / (5.0 * DAY) + (T_soil_K - T_air_K) / (2.0 * DAY))


def main():
    import torch
    ap = argparse.ArgumentParser()
    ap.add_argument("--days", type=float, default=30.0)
    args = ap.parse_args()
    grid, integrator, lat, lon = build()
    land = TerrariumDryLand(integrator)
    lat_d, lon_d = (torch.as_tensor(x, device="cuda", dtype=torch.float32) for x in (lat, lon))
    T_soil = land.initialize()
    T_air = T_soil.clone()
    dt_atm = 900.0                                          # Leapfrog(dt_at_T31 = 15 min)
    t0 = time.perf_counter()
    nsteps = int(args.days * DAY // dt_atm)
    for n in range(nsteps):
        T_air = toy_atmosphere(torch, lat_d, lon_d, n * dt_atm, T_soil, T_air, dt_atm)
        T_soil = land.timestep(T_air, dt_atm)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    print(f"{grid.num_columns} columns, {nsteps} coupling steps ({args.days} days) in {wall:.2f} s; surface soil temperature "
          f"{float(T_soil.min()) - 273.15:.2f} .. {float(T_soil.max()) - 273.15:.2f} degC; status {integrator.state.status()}")


if __name__ == "__main__":
    main()
