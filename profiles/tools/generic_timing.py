"""What the generic boundary kinds cost the per-step launches: the C3 / C4 workloads as they are (branch-free kinds: k_column) and with
the reference's FreeDrainage() at the bottom (Gradient 0 on the pressure head: k_step_wave / k_heun_generic), Euler and Heun.
    python profiles/tools/generic_timing.py"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import numpy as np
import bench
import workloads as W
from terrarium_jl_amd import parallel
out = {}
for wl in ("c3", "c4"):
    for drainage in (False, True):
        w, desc, config, Nz, dt_name = bench.build_workload(W, parallel, wl, 1, 0, "weak")
        if drainage:
            w["bcs"][("pressure_head", "bottom")] = ("gradient", 0.0)
        for heun in (False, True):
            d = W.setup_device(w)
            step = d.step_heun_timed if heun else d.step_timed
            d.step(w["dt"], 10, finalize=False)
            d.save_state()
            ts = []
            for _ in range(7):
                d.restore_state()
                step(w["dt"], 50, finalize=False)
                d.restore_state()
                ts.append(step(w["dt"], 50, finalize=False) * 1e3 / 50)
            out[f"{wl}_{'free_drainage' if drainage else 'as_is'}_{'heun' if heun else 'euler'}"] = round(float(np.median(ts)), 2)
            assert d.status() == 0
            d.close()
print(json.dumps(out), flush=True)
