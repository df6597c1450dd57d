"""A/B of library options on one box, variants interleaved in one process (boxes of the pool differ by +-4 %):
    python profiles/tools/ab_options.py WORKLOAD name:opt=val,opt=val name:opt=val ...  [--steps N] [--reps R] [--shard P] [--heun]
Each variant is one context of the workload with the given trm_set_option values; the timed quantity is the device time of
N steps (HIP events on the context stream, trm_step_timed) and the wall time around the same call.  Prints medians."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import numpy as np
import bench
import workloads as W
from terrarium_jl_amd import parallel

args = sys.argv[1:]
def flag(name, default):
    if name in args:
        i = args.index(name); v = args[i + 1]; del args[i:i + 2]; return int(v)
    return default
steps, reps, shard = flag("--steps", 100), flag("--reps", 7), flag("--shard", 0)
heun = "--heun" in args
if heun:
    args.remove("--heun")
if shard:
    os.environ["TRM_BENCH_SHARD_OF"] = str(shard)
wl, specs = args[0], args[1:]
w, desc, config, Nz, dt_name = bench.build_workload(W, parallel, wl, 1, 0, "weak")
devs = {}
for spec in specs:
    name, _, opts = spec.partition(":")
    d = W.setup_device(w)
    for kv in filter(None, opts.split(",")):
        k, v = kv.split("=")
        d.set_option(k, int(v))
    d.step(w["dt"], 10, finalize=False)
    d.save_state()
    devs[name] = d
res = {k: [] for k in devs}
wall = {k: [] for k in devs}
rng = np.random.default_rng(1)
for rep in range(reps):
    for name in rng.permutation(list(devs)):
        d = devs[name]
        d.restore_state()
        timed = d.step_heun_timed if heun else d.step_timed
        timed(w["dt"], steps, finalize=False)      # untimed: this variant's own clock / cache state
        d.restore_state()
        d.synchronize()
        t0 = time.perf_counter()
        ms = timed(w["dt"], steps, finalize=False)
        d.synchronize()
        wall[name].append((time.perf_counter() - t0) * 1e6 / steps)
        res[name].append(ms * 1e3 / steps)
out = {"workload": wl, "columns": w["Nh"], "steps": steps, "reps": reps,
       "us_per_step": {k: {"median": round(float(np.median(v)), 2), "min": round(min(v), 2), "wall_median": round(float(np.median(wall[k])), 2)} for k, v in res.items()},
       "status": {k: d.status() for k, d in devs.items()}}
print(json.dumps(out), flush=True)
for d in devs.values():
    d.close()
