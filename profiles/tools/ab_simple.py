"""Times the Euler column program (derive off / on) of whatever library TRM_LIBRARY points at; for the diagnostic builds."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import numpy as np
import bench
import workloads as W
from terrarium_jl_amd import parallel
for wl in (sys.argv[1:] or ["c3"]):
    w, desc, config, Nz, dt_name = bench.build_workload(W, parallel, wl, 1, 0, "weak")
    out = []
    for derive in (0, 1):
        d = W.setup_device(w)
        d.set_option("derive_closure_fields", derive)
        d.step(w["dt"], 5, finalize=False)
        d.save_state()
        ts = []
        for rep in range(6):
            d.restore_state()
            d.step(w["dt"], 1, finalize=False)      # (the first step after a restore reads T / liq)
            ts.append(d.step_timed(w["dt"], 150, finalize=False) * 1e3 / 150)
        out.append(f"derive={derive} {np.median(ts):.2f}")
        d.close()
    print(os.environ.get("TRM_LIBRARY", "shipped").split("/")[-1], wl, " ".join(out), "us/step", flush=True)
