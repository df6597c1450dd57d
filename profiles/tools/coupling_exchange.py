"""The per-step LandModel at the size BASELINE config 4 actually runs at: one shard of N145 sharded 8 ways (7 119 columns), its
atmospheric inputs refreshed by a coupled model EVERY step (speedy_dry_land.jl:45-68) -- how the inputs arrive and how the step
is launched.  Wall time per step over K steps (host + device, one wait at the end), median of R repetitions from one state:
   host          trm_set_forcing (host arrays) x 2 + a synchronous trm_step per step -- the round-3 form
   device-copy   trm_set_forcing_device x 2 (stream-ordered D2D) + asynchronous trm_step
   zero-copy     the coupled model writes the library's own input buffers on the library's stream (trm_field_device_ptr,
                 trm_set_stream) + asynchronous trm_step
   zero-copy+1   the same with TRM_OPT_SINGLE_STEP_PROGRAM = 1: the resident column program with the surface processes inline,
                 ONE launch per step instead of the k_surface + k_column pair
   m steps       m = 5, 10, 50 steps between exchanges (the resident multi-step program), inputs written zero-copy
    python profiles/tools/coupling_exchange.py [--shard 8] [--steps 300] [--reps 7] [--workload c4]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import numpy as np
import torch

import bench
import workloads as W
from terrarium_jl_amd import parallel

args = sys.argv[1:]


def flag(name, default, conv=int):
    if name in args:
        i = args.index(name)
        v = args[i + 1]
        del args[i:i + 2]
        return conv(v)
    return default


shard, steps, reps, wl = flag("--shard", 8), flag("--steps", 300), flag("--reps", 7), flag("--workload", "c4", str)
os.environ["TRM_BENCH_SHARD_OF"] = str(shard)
w, desc, config, Nz, dt_name = bench.build_workload(W, parallel, wl, 1, 0, "weak")
Nh, dt = w["Nh"], w["dt"]
stream = torch.cuda.Stream()
rng = np.random.default_rng(2)
Tair = [w["inputs"]["air_temperature"] + 0.01 * k for k in range(8)]
swd = [w["inputs"]["surface_shortwave_down"] + 0.1 * k for k in range(8)]
dT = [torch.as_tensor(x, device="cuda") for x in Tair]
dS = [torch.as_tensor(x, device="cuda") for x in swd]


def make(async_, single=0, spl=1, own_stream=False):
    d = W.setup_device(w, steps_per_launch=spl)
    d.set_option("asynchronous", async_)
    d.set_option("single_step_program", single)
    if own_stream:
        d.set_stream(stream.cuda_stream)
    d.step(dt, 10, finalize=False)
    d.synchronize()
    d.save_state()
    return d


def timed(d, body):
    out = []
    for _ in range(reps):
        d.restore_state()
        d.synchronize()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        body(d)
        d.synchronize()
        out.append((time.perf_counter() - t0) * 1e6 / steps)
    return dict(median=round(float(np.median(out)), 2), min=round(min(out), 2))


def host(d):
    for n in range(steps):
        d.set_forcing("air_temperature", Tair[n % 8])
        d.set_forcing("surface_shortwave_down", swd[n % 8])
        d.step(dt, 1, finalize=False)


def device_copy(d):
    for n in range(steps):
        d.set_forcing_device("air_temperature", dT[n % 8].data_ptr())
        d.set_forcing_device("surface_shortwave_down", dS[n % 8].data_ptr())
        d.step(dt, 1, finalize=False)


def zero_copy(m):
    def body(d):
        vT = torch.as_tensor(d.device_array("air_temperature"), device="cuda")
        vS = torch.as_tensor(d.device_array("surface_shortwave_down"), device="cuda")
        with torch.cuda.stream(stream):
            for n in range(0, steps, m):
                vT.copy_(dT[(n // m) % 8], non_blocking=True)      # (stands for the coupled model's own kernel writing the buffers)
                vS.copy_(dS[(n // m) % 8], non_blocking=True)
                d.step(dt, min(m, steps - n), finalize=False)
    return body


res = {}
res["host"] = timed(make(0), host)
res["device-copy"] = timed(make(1), device_copy)
res["zero-copy"] = timed(make(1, own_stream=True), zero_copy(1))
res["zero-copy+single-step-program"] = timed(make(1, single=1, own_stream=True), zero_copy(1))
for m in (5, 10, 50):
    res[f"zero-copy, {m} steps per exchange"] = timed(make(1, spl=0, own_stream=True), zero_copy(m))
# device time of the bare launches for reference (no exchange): the launch pair, the single-step program
for name, single in (("device: launch pair", 0), ("device: single-step program", 1)):
    d = make(0, single=single)
    ts = []
    for _ in range(reps):
        d.restore_state()
        ms = sum(d.step_timed(dt, 1, finalize=False) for _ in range(50))
        ts.append(ms * 1e3 / 50)
    res[name] = dict(median=round(float(np.median(ts)), 2), min=round(min(ts), 2))
print(json.dumps(dict(workload=wl, shard_of=shard, columns=Nh, steps=steps, reps=reps, us_per_step=res)), flush=True)
