"""How long the HBM-resident step needs to reach its steady state on a fresh box: one context of the workload (default c3x8), the same
60 steps from the same state timed over and over for `seconds` (default 90), printed against the wall clock since the first launch.
(round 4: within ONE gpurun call the memory-only floor of the access pattern fell from 193 to 173 us and the step from 206 to 192 us
between the first and the second minute -- profiles/r04/step_vs_floor.log)
    python profiles/tools/ramp_long.py [workload] [seconds] [idle_seconds_between_blocks]"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import numpy as np
import bench
import workloads as W
from terrarium_jl_amd import parallel
wl = sys.argv[1] if len(sys.argv) > 1 else "c3x8"
seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 90.0
idle = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
w, desc, config, Nz, dt_name = bench.build_workload(W, parallel, wl, 1, 0, "weak")
d = W.setup_device(w)
d.step(w["dt"], 10, finalize=False)
d.save_state()
t0 = time.perf_counter()
rows = []
while time.perf_counter() - t0 < seconds:
    d.restore_state()
    ms = d.step_timed(w["dt"], 60, finalize=False)
    rows.append((round(time.perf_counter() - t0, 2), round(ms * 1e3 / 60, 2)))
    if idle:
        time.sleep(idle)
# one line per ~2 s: the median of the blocks in it
out, k = [], 0
while k < len(rows):
    j = k
    while j < len(rows) and rows[j][0] < rows[k][0] + 2.0:
        j += 1
    out.append((rows[k][0], float(np.median([r[1] for r in rows[k:j]]))))
    k = j
print(json.dumps({"workload": wl, "idle_s": idle, "blocks": len(rows), "t_s__us_per_step": out}))
