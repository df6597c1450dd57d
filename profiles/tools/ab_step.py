"""A/B of the step kernels on one box, interleaved: legacy k_step_wave, the column program with / without the
derivation of T and liq, Heun, multi-step.  Usage: python profiles/tools/ab_step.py [workload ...]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import numpy as np
import bench
import workloads as W
from terrarium_jl_amd import parallel

def variants(dev, name):
    if name == "legacy": dev.set_option("step_kernel", "unfused")
    elif name == "column": dev.set_option("derive_closure_fields", 0)
    elif name == "column+derive": dev.set_option("derive_closure_fields", 1)
    elif name.startswith("multi"): dev.set_option("steps_per_launch", int(name[5:]))

for wl in (sys.argv[1:] or ["c3", "c3x8"]):
    w, desc, config, Nz, dt_name = bench.build_workload(W, parallel, wl, 1, 0, "weak")
    devs = {}
    for name in ("column", "column+derive", "multi10", "multi50"):
        devs[name] = W.setup_device(w)
        variants(devs[name], name)
        devs[name].step(w["dt"], 10, finalize=False)
        devs[name].save_state()
    res = {k: [] for k in devs}
    heun = []
    rng = np.random.default_rng(1)
    for rep in range(7):
        for name in rng.permutation(list(devs)):
            d = devs[name]
            d.restore_state()
            d.step_timed(w["dt"], 150, finalize=False)      # untimed: this variant's own clock / cache state
            d.restore_state()
            res[name].append(d.step_timed(w["dt"], 150, finalize=False) * 1e3 / 150)   # us per step
    for name, d in devs.items():
        d.restore_state()
    import time
    dh = devs["column"]
    for rep in range(4):
        dh.restore_state()
        dh.synchronize()
        t0 = time.perf_counter(); dh.step_heun(w["dt"], 100, finalize=False); dh.synchronize()
        heun.append((time.perf_counter() - t0) * 1e4)
    print(wl, " ".join(f"{k} {np.median(v):.2f} (min {min(v):.2f})" for k, v in res.items()), f"heun {min(heun):.2f}", "us/step", flush=True)
    for d in devs.values():
        d.close()
