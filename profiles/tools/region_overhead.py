"""Where the wall clock of a short timed region goes (bench.py brackets K steps with synchronisations): wall - device time for
K = 1 ... 100 at C3, and the cost of the bracketing calls on an idle stream.  python profiles/tools/region_overhead.py"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import numpy as np
import torch
import bench
import workloads as W
from terrarium_jl_amd import parallel
w, desc, config, Nz, dt_name = bench.build_workload(W, parallel, "c3", 1, 0, "weak")
d = W.setup_device(w)
d.step(w["dt"], 10, finalize=False)
d.save_state()
for _ in range(20):
    d.step_timed(w["dt"], 100, finalize=False); d.restore_state()
out = {}
def med(f, n=200):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); f(); ts.append((time.perf_counter() - t0) * 1e6)
    return round(float(np.median(ts)), 2)
out["idle_torch_synchronize_us"] = med(torch.cuda.synchronize)
out["idle_trm_synchronize_us"] = med(d.synchronize)
out["idle_step_timed_0_steps_us"] = med(lambda: d.step_timed(w["dt"], 0, finalize=False))
for K in (1, 5, 20, 100):
    walls, devs = [], []
    for _ in range(30):
        d.restore_state(); d.synchronize(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        ms = d.step_timed(w["dt"], K, finalize=False)
        d.synchronize(); torch.cuda.synchronize()
        walls.append((time.perf_counter() - t0) * 1e6); devs.append(ms * 1e3)
    out[f"K{K}"] = {"wall_us": round(float(np.median(walls)), 1), "device_us": round(float(np.median(devs)), 1),
                    "overhead_us": round(float(np.median(np.array(walls) - np.array(devs))), 1), "device_us_per_step": round(float(np.median(devs)) / K, 2)}
print(json.dumps(out))
