"""First step at which a synthetic bench workload raises a status flag (the explicit Richards scheme at dt = 60 s dries top
cells out after a few hundred steps, in the reference as well): bounds the stretch bench.py may step from one state.
    python profiles/tools/first_flag.py c3 c4 ..."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import bench
import workloads as W
from terrarium_jl_amd import parallel
for wl in sys.argv[1:]:
    w, desc, config, Nz, dt_name = bench.build_workload(W, parallel, wl, 1, 0, "weak")
    d = W.setup_device(w)
    first = None
    for n in range(0, 400, 10):
        d.step(w["dt"], 10, finalize=False)
        if d.status():
            first = n + 10
            break
    print(json.dumps({"workload": wl, "first_flag_within_steps": first, "flags": d.status()}), flush=True)
    d.close()
