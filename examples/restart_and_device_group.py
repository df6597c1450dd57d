#!/usr/bin/env python3
"""Two things a long simulation needs around the time step, with the host mirror:

1. RESTART (docs/src/running/time_stepping.md:97-139 of the reference: a run is resumed from saved prognostic fields):
   `trm.checkpoint(integrator)` -> a dict of host arrays + the clock; `trm.restore(fresh_integrator, ckpt)` into an integrator that
   a cold `initialize` of the same set-up produced (a new process, another device) -- the continued run equals the uninterrupted
   one bit for bit.

2. ONE HOST THREAD, SEVERAL DEVICES (the reference's host is one Julia process: column_grid.jl:32, model_integrator.jl:72-88):
   the columns block-sharded over the devices (`parallel.shard_range`), one integrator per device, `trm.DeviceGroup` steps them
   all without waiting in between (trm_step_all) and combines global diagnostics inside the library (trm_reduce_global_all).
   On a one-GPU machine the shards share the device: same calls, same numbers.

    python examples/restart_and_device_group.py [shards]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import terrarium_jl_amd as trm  # noqa: E402
from terrarium_jl_amd import parallel  # noqa: E402


def soil_model(num_columns, lo=0, device=0):
    """Heat + Richards with a prescribed surface temperature and free drainage; column i carries its own surface temperature."""
    grid = trm.ColumnGrid(trm.ExponentialSpacing(N=30), num_columns, device=device)
    soil = trm.SoilEnergyWaterCarbon(hydrology=trm.SoilHydrology(vertical_flow=trm.RichardsEq()))
    Ts = 2.0 + 0.01 * (lo + np.arange(num_columns))
    bcs = trm.merge_boundary_conditions(trm.PrescribedSurfaceTemperature("Ts", Ts), trm.FreeDrainage())
    return trm.initialize(trm.SoilModel(grid, soil=soil), trm.ForwardEuler(dt=30.0), boundary_conditions=bcs,
                          initializers=dict(temperature=1.0, saturation_water_ice=lambda x, z: min(1.0, 0.8 - 0.05 * z)))


def main():
    import torch
    nshards = int(sys.argv[1]) if len(sys.argv) > 1 else max(2, torch.cuda.device_count())
    ndev = torch.cuda.device_count()
    Nh = 1000

    # ---- 1. restart ------------------------------------------------------------------------------------------------------
    whole = soil_model(Nh)
    trm.run(whole, steps=60)
    ckpt = trm.checkpoint(whole)                          # (np.savez(path, **ckpt["fields"]) + the clock is a restart file)
    trm.run(whole, steps=40)
    resumed = soil_model(Nh)                              # a cold start of the same set-up; its state is replaced
    trm.restore(resumed, ckpt)
    trm.run(resumed, steps=40)
    same = all(np.array_equal(whole.state.get(n), resumed.state.get(n)) for n in trm.restart_fields(whole))
    print(f"restart: 60 + 40 steps through a checkpoint == 100 steps: {same}; clock {resumed.state.clock()}, status {resumed.state.status()}")

    # ---- 2. one host thread, `nshards` contexts on {ndev} device(s) ---------------------------------------------------------
    shards = []
    for r in range(nshards):
        lo, hi = parallel.shard_range(Nh, nshards, r)
        shards.append(soil_model(hi - lo, lo=lo, device=r % ndev))
    group = trm.DeviceGroup([s.state for s in shards])
    for s in shards:
        s.state.set_option("asynchronous", 1)             # nothing waits between the contexts: each has its own stream
    if nshards <= ndev:
        group.comm_init()                                 # (RCCL: one rank per device; otherwise the library folds on the host)
    group.step(30.0, 60, finalize=True)
    group.step(30.0, 40, finalize=True)          # (as the single context above: two finalizing runs)
    group.synchronize()
    T = group.gather("temperature")
    print(f"device group of {nshards} on {ndev} device(s): gathered == single context: {np.array_equal(T, whole.state.get('temperature'))}; "
          f"global max T {group.reduce_global('temperature', 'max').max():.4f}, status {group.status_global()}")


if __name__ == "__main__":
    main()
