"""ONE host thread driving several contexts ("1 process x 8 HIP devices"; the reference's host is one Julia
process, column_grid.jl:32, model_integrator.jl:72-88): the block-sharded N145 columns in three asynchronous contexts on the
box's GPU, stepped with trm_step_all, diagnostics through trm_reduce_global_all / trm_status_global_all -- gathered fields
bit-equal to the unsharded run.  RCCL takes one rank per device: the grouped communicator set-up (trm_comm_init_all) runs with
every context on a device of its own where the box has more than one GPU, with one context otherwise."""
import ctypes as C

import numpy as np
import pytest

import terrarium_jl_amd as trm
import workloads as W
from terrarium_jl_amd import parallel

pytestmark = pytest.mark.gpu


shard_workload = W.shard_workload


@pytest.mark.parametrize("config,heun", [("richards", False), ("land", False), ("richards", True)])
def test_three_contexts_from_one_thread_equal_the_unsharded_run(config, heun):
    lat, lon = W.columns_from_mask("N145")
    w = W.make_workload(config, lat, lon, 32)
    whole = W.setup_device(w, steps_per_launch=0)
    shards = [W.setup_device(shard_workload(w, *parallel.shard_range(w["Nh"], 3, r)), steps_per_launch=0) for r in range(3)]
    assert sum(s.grid.Nh for s in shards) == w["Nh"]
    for s in shards:
        s.set_option("asynchronous", 1)          # nothing waits between the contexts: each has its own stream
    group = trm.DeviceGroup(shards)
    for n, fin in ((7, False), (1, False), (12, True)):
        (whole.step_heun if heun else whole.step)(w["dt"], n, finalize=fin)
        group.step(w["dt"], n, finalize=fin, heun=heun)
    group.synchronize()
    assert all(s.clock() == whole.clock() for s in shards)
    for name in W.compared_fields(w):
        assert np.array_equal(group.gather(name), np.atleast_2d(whole.get(name)), equal_nan=True), name
    # global diagnostics from one thread: exact for min / max / hasnan, sums to rounding (the fold order differs)
    for name in ("temperature", "saturation_water_ice"):
        for op in ("min", "max", "hasnan"):
            assert np.array_equal(group.reduce_global(name, op), whole.reduce(name, op)), (name, op)
        assert np.allclose(group.reduce_global(name, "sum"), whole.reduce(name, "sum"), rtol=1e-12, atol=0)
    assert np.isclose(group.reduce_global("saturation_water_ice", "volume_integral_z")[0], whole.reduce("saturation_water_ice", "volume_integral_z")[0], rtol=1e-12)
    assert group.status_global() == whole.status() == 0
    # a NaN in ONE shard reaches the global minimum / maximum and the status word
    T = shards[1].get("temperature")
    T[3, 5] = np.nan
    shards[1].set("temperature", T)
    assert np.isnan(group.reduce_global("temperature", "min")[3]) and np.isnan(group.reduce_global("temperature", "max")[3])
    assert group.reduce_global("temperature", "hasnan")[3] == 1 and group.reduce_global("temperature", "hasnan")[2] == 0
    group.step(w["dt"], 1)
    group.synchronize()
    assert group.status_global() & 1


def test_grouped_communicator_setup():
    """trm_comm_init_all: every context on a device of its own (all GPUs of the box; one on a one-GPU box) -- the grouped
    ncclCommInitRank, then the grouped all-reduces behind trm_reduce_global_all; two contexts on ONE device are refused with a
    message instead of a hang."""
    import torch
    ndev = torch.cuda.device_count()
    lat, lon = W.columns_from_mask("N72")
    w = W.make_workload("richards", lat[:1200], lon[:1200], 20)
    shards = [W.setup_device(shard_workload(w, *parallel.shard_range(w["Nh"], ndev, r)), device=r) for r in range(ndev)]
    whole = W.setup_device(w)
    group = trm.DeviceGroup(shards)
    group.comm_init()
    assert all(s.comm_world() == ndev for s in shards)
    for d in shards + [whole]:
        d.step(w["dt"], 5, finalize=True)
    for op in ("min", "max", "hasnan"):
        assert np.array_equal(group.reduce_global("temperature", op), whole.reduce("temperature", op)), op
    assert np.allclose(group.reduce_global("temperature", "sum"), whole.reduce("temperature", "sum"), rtol=1e-12, atol=0)
    assert group.status_global() == 0
    two = [W.setup_device(shard_workload(w, 0, 600)), W.setup_device(shard_workload(w, 600, 1200))]
    with pytest.raises(trm.TerrariumHipError, match="two contexts on one device"):
        trm.DeviceGroup(two).comm_init()
    for d in two:
        d.step(w["dt"], 5, finalize=True)
    assert np.array_equal(trm.DeviceGroup(two).reduce_global("temperature", "max"), whole.reduce("temperature", "max"))   # (the host fold)


def test_inputs_from_device_memory_between_asynchronous_steps():
    """trm_set_forcing_device: the coupled atmosphere's fields arrive from device memory, stream-ordered, between asynchronous
    single steps (speedy_dry_land.jl:45-68) -- same results as host uploads with a synchronous context; the single-step
    program (TRM_OPT_SINGLE_STEP_PROGRAM: one launch per step instead of the launch pair) gives the same bits."""
    import torch
    lat, lon = W.synthetic_columns(700)
    w = W.make_workload("land", lat, lon, 32)
    a, b, c = W.setup_device(w), W.setup_device(w), W.setup_device(w)
    b.set_option("asynchronous", 1)
    c.set_option("asynchronous", 1)
    c.set_option("single_step_program", 1)
    dev_T = torch.empty(w["Nh"], dtype=torch.float64, device="cuda")
    dev_sw = torch.empty(w["Nh"], dtype=torch.float64, device="cuda")
    for n in range(12):
        phase = 2 * np.pi * (n * w["dt"]) / 86400.0 - w["lon"]
        Tair, sw = w["T0"] + 5.0 * np.sin(phase) + 0.1 * n, np.maximum(0.0, 600.0 * np.sin(phase))
        a.set_forcing("air_temperature", Tair)
        a.set_forcing("surface_shortwave_down", sw)
        a.step(w["dt"], 1, finalize=(n == 11))
        for d in (b, c):
            d.synchronize()                       # (the staging tensors are reused: the previous copies have executed)
            dev_T.copy_(torch.as_tensor(Tair)); dev_sw.copy_(torch.as_tensor(sw))
            torch.cuda.synchronize()
            d.set_forcing_device("air_temperature", dev_T.data_ptr())
            d.set_forcing_device("surface_shortwave_down", dev_sw.data_ptr())
            d.step(w["dt"], 1, finalize=(n == 11))
    for d in (b, c):
        d.synchronize()
        for name in W.compared_fields(w):
            assert np.array_equal(a.get(name), d.get(name), equal_nan=True), name
        assert d.clock() == a.clock() and d.status() == a.status()
