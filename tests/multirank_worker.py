"""One rank of the multi-rank GPU test (tests/test_gpu_multirank.py): a fresh process that owns the context of its block of
the N145 columns on the one GPU of the box, steps it, and takes part in the global diagnostics over gloo.  Rank 0 also
steps the unsharded grid and checks that sharding changed nothing."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]

import numpy as np
import torch.distributed as dist

import workloads as W
import terrarium_jl_amd as trm
from terrarium_jl_amd import parallel


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    config, nsteps = sys.argv[1], int(sys.argv[2])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lat, lon = W.columns_from_mask("N145")
    full = W.make_workload(config, lat, lon, 32, hydraulics="default")
    lo, hi = parallel.shard_range(lat.size, world, rank)
    w = dict(full)
    w["Nh"] = hi - lo
    w["fields"] = {k: (v[..., lo:hi] if np.ndim(v) else v) for k, v in full["fields"].items()}
    w["bcs"] = {k: (kind, (val[lo:hi] if np.ndim(val) else val)) for k, (kind, val) in full["bcs"].items()}
    w["inputs"] = {k: (v[lo:hi] if np.ndim(v) else v) for k, v in full["inputs"].items()}
    # the shard keeps the x spacing of the global grid (dx enters the flux boundary terms as Az / V)
    p = trm._capi.default_params()
    for k, v in w["params"].items():
        setattr(p, k, v)
    grid = trm.ColumnGrid(trm.PrescribedSpacing(dz=list(w["thickness"])), w["Nh"])
    grid.dx = 1.0 / lat.size
    dev = trm.DeviceState(grid, p)
    for name, v in w["fields"].items():
        dev.set(name, v)
    for (var, side), (kind, value) in w["bcs"].items():
        dev.set_bc(var, side, kind, value)
    for name, v in w["inputs"].items():
        dev.set_forcing(name, v)
    dev.initialize()
    dev.step(w["dt"], nsteps, finalize=True)
    if rank == 1:
        T = dev.get("temperature")
        T[5, 3] = np.nan            # a NaN on one rank must surface in the global HASNAN / MIN / MAX
        dev.set("temperature", T)
    names = W.compared_fields(full)
    gathered = {n: parallel.gather_columns(dev.get(n), lat.size) for n in names}
    red = {(n, op): parallel.global_reduce(dev, n, op) for n in ("internal_energy", "saturation_water_ice", "temperature")
           for op in ("sum", "min", "max", "hasnan")}
    vol = parallel.global_reduce(dev, "saturation_water_ice", "volume_integral_z")
    status = parallel.global_status(dev.status())
    if rank == 0:
        ref = W.setup_device(full)
        ref.step(full["dt"], nsteps, finalize=True)
        for n in names:
            a, b = gathered[n], ref.get(n)
            if n == "temperature":
                assert np.isnan(a[5, parallel.shard_range(lat.size, world, 1)[0] + 3])
                a = a.copy(); a[5, parallel.shard_range(lat.size, world, 1)[0] + 3] = b[5, parallel.shard_range(lat.size, world, 1)[0] + 3]
            assert np.array_equal(a, b), n          # sharding changes nothing, bit for bit
        for n in ("internal_energy", "saturation_water_ice"):
            f = ref.get(n)
            assert np.array_equal(red[(n, "min")], f.min(axis=1)) and np.array_equal(red[(n, "max")], f.max(axis=1)), n
            assert np.allclose(red[(n, "sum")], f.sum(axis=1), rtol=1e-12, atol=0), n
            assert np.all(red[(n, "hasnan")] == 0)
        assert red[("temperature", "hasnan")][5] == 1 and red[("temperature", "hasnan")].sum() == 1
        # a NaN on one rank reaches the global minimum / maximum (Base.minimum semantics; parallel.combine carries a flag)
        assert np.isnan(red[("temperature", "min")][5]) and np.isnan(red[("temperature", "max")][5])
        keep = np.arange(red[("temperature", "min")].size) != 5
        assert np.array_equal(red[("temperature", "min")][keep], ref.get("temperature").min(axis=1)[keep])
        assert np.isclose(vol[0], ref.reduce("saturation_water_ice", "volume_integral_z")[0], rtol=1e-12)
        assert status == ref.status() == 0, (status, ref.status(), dev.status())
        print("multirank ok", config, world, "ranks")
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
