"""Feasibility probe: the columns of a LandModel workload split over two contexts (own streams, asynchronous calls), stepped
in interleaved batches, against one context with all columns.  python profiles/tools/two_stream_probe.py [config] [hydraulics] [batch]"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import terrarium_jl_amd as trm  # noqa: E402
import workloads as W  # noqa: E402

config = sys.argv[1] if len(sys.argv) > 1 else "land"
hyd = sys.argv[2] if len(sys.argv) > 2 else "default"
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 1
lat, lon = W.columns_from_mask("N145")
half = lat.size // 2
full = W.setup_device(W.make_workload(config, lat, lon, 32, hydraulics=hyd))
A = W.setup_device(W.make_workload(config, lat[:half], lon[:half], 32, hydraulics=hyd))
B = W.setup_device(W.make_workload(config, lat[half:], lon[half:], 32, hydraulics=hyd))
dt = 0.05 if config == "landveg" else 60.0
for d in (full, A, B):
    d.step(dt, 20, finalize=False)
    d.save_state()
    d.set_option("asynchronous", 1)
steps = 120
out = {}
def run_full():
    full.restore_state(); full.synchronize()
    t0 = time.perf_counter()
    full.step(dt, steps, finalize=False)
    full.synchronize()
    return (time.perf_counter() - t0) / steps * 1e6
def run_split(first_alone):
    A.restore_state(); B.restore_state(); A.synchronize(); B.synchronize()
    t0 = time.perf_counter()
    if first_alone:
        A.step(dt, 1, finalize=False)       # a head start for A: its surface launches then fall into B's column launches
    for n in range(0, steps, batch):
        if not (first_alone and n == 0):
            A.step(dt, batch, finalize=False)
        elif batch > 1:
            A.step(dt, batch - 1, finalize=False)
        B.step(dt, batch, finalize=False)
    A.synchronize(); B.synchronize()
    return (time.perf_counter() - t0) / steps * 1e6
out["full_us_per_step"] = round(min(run_full() for _ in range(5)), 2)
out["split_us_per_step"] = round(min(run_split(False) for _ in range(5)), 2)
out["split_skewed_us_per_step"] = round(min(run_split(True) for _ in range(5)), 2)
print(json.dumps(out))
