# round 3, GPU call 9: Euler column program looping over 2 / 4 / 8 workgroup chunks per workgroup (fewer, longer-lived waves)
run() { local limit=$1; shift; timeout -k 10 $limit "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: $*"; exit 1; fi; return 0; }
L=gpurun_out/exp9_ab.log; : > $L
AB="python profiles/tools/ab_options.py"
run 300 python -m pytest tests/test_gpu_column_programs.py -q -x -k "euler_program" -W ignore::DeprecationWarning > gpurun_out/exp9_tests.log 2>&1; tail -2 gpurun_out/exp9_tests.log
run 300 $AB c3x8 g1:chunks=1 g2:chunks=2 g4:chunks=4 g8:chunks=8 --steps 60 --reps 5 >> $L 2>&1
run 300 $AB c3 g1:chunks=1 g2:chunks=2 g4:chunks=4 g8:chunks=8 >> $L 2>&1
cat $L
python - <<'PY'
import sys, os
sys.path[:0] = [os.getcwd(), "tests", "oracle"]
import numpy as np, workloads as W
lat, lon = W.columns_from_mask("N72")
w = W.make_workload("richards", lat[:1003], lon[:1003], 32)
a, b = W.setup_device(w), W.setup_device(w)
a.set_option("chunks", 4)
for d in (a, b):
    d.set_option("derive_closure_fields", 1)
    d.step(w["dt"], 25, finalize=True)
print("bitwise", all(np.array_equal(a.get(n), b.get(n)) for n in W.compared_fields(w)), a.status(), b.status())
PY
