import sys, os, time
sys.path[:0] = [os.getcwd(), 'tests', 'oracle']
import numpy as np, workloads as W, torch
lat, lon = W.columns_from_mask("N145")
def run(nsplit, steps=100, reps=3):
    devs = []
    for s in range(nsplit):
        lo, hi = s * lat.size // nsplit, (s + 1) * lat.size // nsplit
        w = W.make_workload("richards", lat[lo:hi], lon[lo:hi], 32)
        d = W.setup_device(w); d.set_option("asynchronous", 1); devs.append((d, w))
    for d, w in devs: d.step(w["dt"], 10, False)
    for d, w in devs: d.synchronize()
    best = 1e9
    for r in range(reps):
        for d, w in devs: d.restore_state() if r else d.save_state()
        for d, w in devs: d.synchronize()
        t0 = time.perf_counter()
        # interleave the enqueues so that neither stream runs ahead by the whole loop
        for n in range(steps // 10):
            for d, w in devs: d.step(w["dt"], 10, False)
        for d, w in devs: d.synchronize()
        best = min(best, (time.perf_counter() - t0) / steps)
    print(f"{nsplit} contexts / streams: {best * 1e6:.2f} us per step of all {lat.size} columns -> {lat.size / best / 1e9:.3f} Gcs/s")
for n in (1, 2, 3, 4, 8):
    run(n)
