"""A/B of a library build on the deep-column step (Nz = 100, N145 columns: heat-only and heat + Richards, Euler and Heun; land at
14 017 columns): one process per build (TRM_LIBRARY), prints medians.  python profiles/tools/deep_ab.py NAME"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import numpy as np
import workloads as W
name = sys.argv[1] if len(sys.argv) > 1 else "shipped"
out = {"build": name}
for config, mask, Nz in (("heat", "N145", 100), ("richards", "N145", 100), ("land", "N72", 100), ("richards", "N145", 128)):
    lat, lon = W.columns_from_mask(mask)
    w = W.make_workload(config, lat, lon, Nz)
    for heun in (False, True):
        d = W.setup_device(w)
        step = d.step_heun_timed if heun else d.step_timed
        d.step(w["dt"], 10, finalize=False)
        d.save_state()
        ts = []
        for _ in range(5):
            d.restore_state()
            step(w["dt"], 40, finalize=False)
            d.restore_state()
            ts.append(step(w["dt"], 40, finalize=False) * 1e3 / 40)
        out[f"{config}_{mask}_Nz{Nz}_{'heun' if heun else 'euler'}"] = round(float(np.median(ts)), 2)
        assert d.status() == 0
        d.close()
print(json.dumps(out), flush=True)
