import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import numpy as np
import workloads as W
from terrarium_jl_amd import parallel
lat, lon = W.columns_from_mask("N145")
for world, rank in ((2, 0), (2, 1), (1, 0)):
    lo, hi = parallel.shard_range(lat.size, world, rank)
    w = W.make_workload("land", lat[lo:hi], lon[lo:hi], 32)
    res = {}
    for name, opts in (("per_step", dict(steps_per_launch=1)), ("per_step_noderive", dict(steps_per_launch=1, derive_closure_fields=0)), ("multi", dict(steps_per_launch=0))):
        d = W.setup_device(w)
        for k, v in opts.items():
            d.set_option(k, v)
        first = None
        for n in range(0, 100, 10):
            d.step(w["dt"], 10, finalize=False)
            if d.status() and first is None:
                first = n + 10
        res[name] = (first, d.status(), float(np.nanmax(np.abs(d.get("temperature")))))
        d.close()
    print(world, rank, hi - lo, res, flush=True)
