import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import numpy as np
import torch
import bench
import workloads as W
from terrarium_jl_amd import parallel
sync = torch.cuda.synchronize
for rank in (0, 1):
    w, desc, config, Nz, dt_name = bench.build_workload(W, parallel, "c4", 2, rank, "strong")
    for spl in (1, 0):
        dev = W.setup_device(w)
        dev.set_option("step_kernel", "fused")
        dev.set_option("steps_per_launch", spl)
        m = bench.measure(dev, w, config, 50, 10, 100.0, False, sync, None, 5)
        print(rank, w["Nh"], spl, round(m["kernel_us_per_step"], 2), dev.status(), flush=True)
        dev.close()
