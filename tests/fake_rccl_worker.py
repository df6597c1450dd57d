"""Worker of tests/test_gpu_fake_rccl.py: a FRESH process (the library resolves its collective library once, on first use) in
which libterrarium_hip.so opens tests/fake_rccl.c instead of librccl.so, so that the grouped call sequences of
trm_comm_init_all / trm_reduce_global_all / trm_status_global_all run with EIGHT communicators on the one GPU of the box.
Validates sequencing and packing only -- no transport, no topology, no scaling: nothing here is a measurement."""
import ctypes as C
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]


def build_fake():
    out = os.path.join(tempfile.mkdtemp(prefix="fake_rccl_"), "libfake_rccl.so")
    subprocess.check_call(["gcc", "-shared", "-fPIC", "-O1", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", os.path.join(ROOT, "tests", "fake_rccl.c"),
                           "-L/opt/rocm/lib", "-lamdhip64", "-o", out])
    return out


def main():
    fake = build_fake()
    os.environ["TRM_RCCL_LIBRARY"] = fake
    os.environ["TRM_RCCL_ALLOW_SHARED_DEVICE"] = "1"
    import numpy as np
    import terrarium_jl_amd as trm
    import workloads as W
    from terrarium_jl_amd import parallel

    res = {}
    n = 8
    lat, lon = W.columns_from_mask("N72")
    w = W.make_workload("land", lat[:4000], lon[:4000], 20)

    def shards():
        return [W.setup_device(W.shard_workload(w, *parallel.shard_range(w["Nh"], n, r))) for r in range(n)]

    a, b = shards(), shards()          # a: communicators (the fake), b: none (the library's host fold)
    ga, gb = trm.DeviceGroup(a), trm.DeviceGroup(b)
    fake_lib = C.CDLL(fake)            # the same mapping the library opened: its counters
    ga.comm_init()
    res["world"] = [s.comm_world() for s in a]
    res["rank"] = [s.comm_rank() for s in a]
    res["live_after_init"] = fake_lib.fake_rccl_live_communicators()
    for g in (ga, gb):
        g.step(w["dt"], 6, finalize=True)
        g.synchronize()
    same = True
    checked = 0
    for name in ("temperature", "saturation_water_ice", "ground_heat_flux", "hydraulic_conductivity"):
        for op in ("sum", "min", "max", "hasnan"):
            x, y = ga.reduce_global(name, op), gb.reduce_global(name, op)
            same = same and np.array_equal(x, y, equal_nan=True)
            checked += 1
    x, y = ga.reduce_global("saturation_water_ice", "volume_integral_z"), gb.reduce_global("saturation_water_ice", "volume_integral_z")
    same = same and np.array_equal(x, y)
    res["reductions_equal_host_fold_bitwise"] = bool(same)
    res["reductions_checked"] = checked + 1
    res["status_clean"] = [ga.status_global(), gb.status_global()]
    # a NaN planted in ONE shard reaches every reduction and the status word the same way on both paths
    for g in (a, b):
        T = g[5].get("temperature")
        T[3, 7] = np.nan
        g[5].set("temperature", T)
    nan_same = True
    for op in ("min", "max", "hasnan", "sum"):
        x, y = ga.reduce_global("temperature", op), gb.reduce_global("temperature", op)
        nan_same = nan_same and np.array_equal(x, y, equal_nan=True)
        if op in ("min", "max"):
            nan_same = nan_same and bool(np.isnan(x[3])) and not np.isnan(x[2])
        if op == "hasnan":
            nan_same = nan_same and x[3] == 1 and x[2] == 0
    res["nan_reaches_every_rank"] = bool(nan_same)
    for g in (ga, gb):
        g.step(w["dt"], 1, finalize=False)
        g.synchronize()
    res["status_after_nan"] = [ga.status_global(), gb.status_global()]
    # a list that is not ONE group in rank order is refused instead of posted (real RCCL would wait in ncclGroupEnd)
    c = shards()
    gc = trm.DeviceGroup(c)
    gc.comm_init()                                   # a second group of eight
    refused = {}
    for label, lst in (("subset", a[:7]), ("mixed_groups", a[:7] + [c[7]]), ("out_of_order", a[1:] + a[:1]), ("some_without", a[:7] + [b[7]])):
        try:
            trm.DeviceGroup(lst).reduce_global("temperature", "max")
            refused[label] = "accepted"
        except trm.TerrariumHipError as e:
            refused[label] = "refused" if "do not form ONE group" in str(e) else f"other: {e}"
        try:
            trm.DeviceGroup(lst).status_global()
            refused[label] += "/accepted"
        except trm.TerrariumHipError:
            refused[label] += "/refused"
    res["refused"] = refused
    res["group_c_works"] = bool(np.array_equal(gc.reduce_global("temperature", "max"), trm.DeviceGroup(shards()).reduce_global("temperature", "max")))
    # a context alone: trm_comm_init(world = 1) / trm_reduce_global through the same entry points
    one = W.setup_device(w)
    one.comm_init(0, 1, one.comm_unique_id())
    one.step(w["dt"], 2, finalize=True)
    res["single_rank"] = bool(np.array_equal(one.reduce_global("temperature", "max"), one.reduce("temperature", "max")) and one.status_global() == 0)
    # a failing ncclCommInitRank inside the group: no context keeps a communicator, the error names the call, a second attempt works
    d = shards()
    gd = trm.DeviceGroup(d)
    os.environ["FAKE_RCCL_FAIL_INIT_RANK"] = "5"
    try:
        gd.comm_init()
        res["init_failure"] = "accepted"
    except trm.TerrariumHipError as e:
        res["init_failure"] = "refused" if "ncclCommInitRank" in str(e) else f"other: {e}"
    res["world_after_failure"] = [s.comm_world() for s in d]
    del os.environ["FAKE_RCCL_FAIL_INIT_RANK"]
    live_before = fake_lib.fake_rccl_live_communicators()
    gd.comm_init()
    res["world_after_retry"] = [s.comm_world() for s in d]
    res["live_delta_retry"] = fake_lib.fake_rccl_live_communicators() - live_before
    for s in a + c + d:
        s.comm_destroy()
    one.comm_destroy()
    res["live_after_destroy"] = fake_lib.fake_rccl_live_communicators()
    print("RESULT " + json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
