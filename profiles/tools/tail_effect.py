"""Does the last, partly filled round of workgroups cost the C3 step?  ns per column-step against the column count
(8 waves per SIMD x 1024 SIMDs x 2 columns per wave = 16384 columns per full round).
Usage: python profiles/tools/tail_effect.py"""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import terrarium_jl_amd as trm  # noqa: E402
import workloads as W  # noqa: E402

lat0, lon0 = W.columns_from_mask("N145")
out = []
for Nh in (32768, 40960, 49152, 53248, 56951, 57344, 61440, 65536, 73728, 81920, 98304):
    reps = (Nh + lat0.size - 1) // lat0.size
    lat, lon = np.tile(lat0, reps)[:Nh], np.tile(lon0, reps)[:Nh]
    w = W.make_workload("richards", lat, lon, 32)
    d = W.setup_device(w)
    d.step(w["dt"], 20, finalize=False)
    d.save_state()
    best = 1e30
    for _ in range(7):
        d.restore_state()
        best = min(best, d.step_timed(w["dt"], 100, finalize=False) / 100 * 1e3)
    out.append((Nh, round(Nh / 16384, 2), round(best, 2), round(best * 1e3 / Nh, 4)))
    print(out[-1], flush=True)
