import sys, os, time
sys.path[:0] = [os.getcwd(), 'tests', 'oracle']
import numpy as np, workloads as W
lat, lon = W.columns_from_mask("N145")
w = W.make_workload("richards", lat, lon, 32)
d = W.setup_device(w)
d.step(w["dt"], 10, False)
d.save_state()
out = []
for c in range(30):
    d.restore_state()
    ms = d.step_timed(w["dt"], 100, False)
    out.append(round(ms * 10, 2))   # us per step
print("us/step per 100-step chunk (same state each time):", out)
