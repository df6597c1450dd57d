"""Steps one workload with one step-kernel variant (TRM_AB_VARIANT = legacy | column | derive | multiN); for counter runs."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import bench
import workloads as W
from terrarium_jl_amd import parallel
w, desc, config, Nz, dt_name = bench.build_workload(W, parallel, sys.argv[1] if len(sys.argv) > 1 else "c3", 1, 0, "weak")
d = W.setup_device(w)
v = os.environ.get("TRM_AB_VARIANT", "column")
if v == "legacy": d.set_option("step_kernel", "unfused")
elif v == "derive": d.set_option("derive_closure_fields", 1)
elif v == "column": d.set_option("derive_closure_fields", 0)
elif v.startswith("multi"): d.set_option("steps_per_launch", int(v[5:]))
d.step(w["dt"], 1, finalize=False)
d.step(w["dt"], int(os.environ.get("TRM_AB_STEPS", "50")), finalize=False)
print(d.status())
