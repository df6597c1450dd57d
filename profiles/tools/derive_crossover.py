"""Where does deriving T / liq from (U, sat) start to pay?  Euler column program, fp64, derive off / on against the column
count (interleaved, warm run first, median of 7).  python profiles/tools/derive_crossover.py [config] [hydraulics]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import workloads as W  # noqa: E402

config = sys.argv[1] if len(sys.argv) > 1 else "richards"
hyd = sys.argv[2] if len(sys.argv) > 2 else "default"
lat0, lon0 = W.columns_from_mask("N145")
rng = np.random.default_rng(2)
for Nh in (8192, 14017, 20480, 28672, 36864, 45056, 56951, 81920):
    reps = (Nh + lat0.size - 1) // lat0.size
    lat, lon = np.tile(lat0, reps)[:Nh], np.tile(lon0, reps)[:Nh]
    w = W.make_workload(config, lat, lon, 32, hydraulics=hyd)
    devs = {}
    for derive in (0, 1):
        d = W.setup_device(w)
        d.set_option("derive_closure_fields", derive)
        d.step(w["dt"], 10, finalize=False)
        d.save_state()
        devs[derive] = d
    res = {0: [], 1: []}
    for rep in range(7):
        for derive in rng.permutation([0, 1]):
            d = devs[int(derive)]
            d.restore_state(); d.step_timed(w["dt"], 100, finalize=False)
            d.restore_state(); res[int(derive)].append(d.step_timed(w["dt"], 100, finalize=False) * 10.0)
    a, b = np.median(res[0]), np.median(res[1])
    print(config, hyd, Nh, f"stored {a:.2f} derived {b:.2f} us/step  ({(b / a - 1) * 100:+.1f} %)", flush=True)
    for d in devs.values():
        d.close()
