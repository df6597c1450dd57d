"""N > 1 path on CPU: two processes over gloo exercise the column sharding and the
global diagnostic reductions (the only collectives the path has)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, Nh, result_dir):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
    import torch.distributed as dist
    import terrarium_jl_amd as trm
    from terrarium_jl_amd import parallel
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(3)
    full = rng.normal(size=(5, Nh))                     # the "global" field, identical on every rank
    full[3, Nh - 1] = np.nan                            # a NaN in the LAST rank's block only: row 3 of min / max must be NaN everywhere
    local = parallel.shard_columns(full, world, rank)  # this rank's block of columns

    class FakeState:  # stands in for DeviceState.reduce (which needs a GPU): per-rank partials
        def reduce(self, field, op):
            if op == "sum":
                return local.sum(axis=1)
            if op == "min":
                return local.min(axis=1)        # (numpy's min / max propagate NaN like Base.minimum / maximum and trm_reduce)
            if op == "max":
                return local.max(axis=1)
            if op == "hasnan":
                return np.isnan(local).any(axis=1).astype(np.float64)
            return np.array([local.sum()])

    st = FakeState()
    out = dict(
        sum=parallel.global_reduce(st, "temperature", "sum"),
        min=parallel.global_reduce(st, "temperature", "min"),
        max=parallel.global_reduce(st, "temperature", "max"),
        hasnan=parallel.global_reduce(st, "temperature", "hasnan"),
        vol=parallel.global_reduce(st, "temperature", "volume_integral_z"),
        status=np.array([parallel.global_status(2 if rank == 1 else 0)]),
        gathered=parallel.gather_columns(local, Nh),
        full=full,
    )
    np.savez(os.path.join(result_dir, f"rank{rank}.npz"), **out)
    dist.destroy_process_group()


@pytest.mark.parametrize("Nh", [56951, 7])
def test_sharded_reductions_world2(tmp_path, Nh):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), Nh, str(tmp_path)), nprocs=world, join=True)
    for rank in range(world):
        r = np.load(tmp_path / f"rank{rank}.npz")
        full = r["full"]
        assert np.allclose(r["sum"], full.sum(axis=1), rtol=1e-12, equal_nan=True)
        # a NaN held by one rank reaches every rank's global minimum / maximum (parallel.combine's sentinel + flag scheme, the
        # same packing trm_reduce_global uses around its RCCL all-reduce)
        assert np.isnan(r["min"][3]) and np.isnan(r["max"][3])
        assert np.array_equal(r["min"], full.min(axis=1), equal_nan=True)
        assert np.array_equal(r["max"], full.max(axis=1), equal_nan=True)
        assert list(r["hasnan"]) == [0, 0, 0, 1, 0]
        assert np.isnan(r["vol"]).all()
        assert int(r["status"][0]) == 2          # rank 1's flag reaches everyone
        assert np.array_equal(r["gathered"], full, equal_nan=True)


def test_nan_flag_packing_of_the_global_min_max():
    """Host-side unit test of the packing trm_reduce_global and parallel.combine share: NaN -> (neutral element, flag)."""
    sys.path[:0] = [ROOT]
    from terrarium_jl_amd import parallel
    a = np.array([1.0, np.nan, -3.0, np.inf])
    b = np.array([0.5, 2.0, np.nan, 4.0])
    for op, red in (("min", np.minimum), ("max", np.maximum)):
        pa, _ = parallel.pack_nan_flags(a, op)
        pb, _ = parallel.pack_nan_flags(b, op)
        assert not np.isnan(pa).any() and not np.isnan(pb).any()          # nothing a backend's MIN / MAX could drop
        out = parallel.unpack_nan_flags(red(pa, pb))
        expect = np.array([red(1.0, 0.5), np.nan, np.nan, red(np.inf, 4.0)])
        assert np.array_equal(out, expect, equal_nan=True)


def test_shard_ranges_cover_all_columns():
    sys.path[:0] = [ROOT]
    from terrarium_jl_amd import parallel
    for Nh, P in ((56951, 8), (14017, 8), (7, 8), (64, 2), (1, 4)):
        ranges = [parallel.shard_range(Nh, P, r) for r in range(P)]
        assert ranges[0][0] == 0 and ranges[-1][1] == Nh
        assert all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
        assert sum(hi - lo for lo, hi in ranges) == Nh
    assert parallel.shard_range(56951, 8, 0) == (0, 7119) and parallel.shard_range(56951, 8, 7) == (49833, 56951)
