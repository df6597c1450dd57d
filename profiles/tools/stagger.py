"""Experiment (diagnostic build -DTRM_EXP_STAGGER): delay a subset of the first-generation workgroups of the Euler
column program so that two populations of waves alternate their load and compute phases.
    TRM_LIBRARY=build/diag/libtrm_stagger.so python profiles/tools/stagger.py"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import numpy as np
import bench
import workloads as W
from terrarium_jl_amd import parallel

w, desc, config, Nz, dt_name = bench.build_workload(W, parallel, sys.argv[1] if len(sys.argv) > 1 else "c3", 1, 0, "weak")
d = W.setup_device(w)
d.set_option("derive_closure_fields", int(os.environ.get("DERIVE", "0")))
d.step(w["dt"], 10, finalize=False)
d.save_state()
def run(code):
    os.environ["TRM_EXP_STAGGER"] = str(code)
    best = 1e9
    for rep in range(4):
        d.restore_state()
        best = min(best, d.step_timed(w["dt"], 100, finalize=False) * 10.0)
    return best
for _ in range(3): run(0)
print("baseline", run(0), flush=True)
for first_gen in (1792, 3584, 896):
    for shift in (0, 1, 3, 5, 8, 9):
        row = []
        for sleep in (2, 4, 6, 8, 12):
            code = (sleep << 24) | (shift << 20) | (first_gen // 64)
            row.append(f"{run(code):.2f}")
        print(f"first_gen {first_gen} shift {shift} sleep(2,4,6,8,12 x 1024 cyc):", " ".join(row), " | baseline again", f"{run(0):.2f}", flush=True)
