# round 3, GPU call 25: the per-column INPUTS of the column program staged as well (one fetch per array and workgroup through LDS)
# against the previous build (outputs staged only).  Full suite with the rule and with staging forced everywhere first
run() { local limit=$1; shift; timeout -k 10 $limit "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: $*"; exit 1; fi; return 0; }
run 900 python -m pytest tests/test_gpu_column_programs.py tests/test_gpu_parity.py -m gpu -q -x -W ignore::DeprecationWarning > gpurun_out/exp25_tests_auto.log 2>&1; tail -2 gpurun_out/exp25_tests_auto.log
TRM_STAGED_SMALL=1 run 900 python -m pytest tests/test_gpu_column_programs.py tests/test_gpu_parity.py tests/test_gpu_reference_tests.py -m gpu -q -x -W ignore::DeprecationWarning > gpurun_out/exp25_tests_forced.log 2>&1; tail -2 gpurun_out/exp25_tests_forced.log
L=gpurun_out/exp25_staged_inputs.log; : > $L
AB="python profiles/tools/ab_options.py"
for round in 1 2 3; do
  for B in new prev; do
    if [ $B = new ]; then unset TRM_LIBRARY; else export TRM_LIBRARY=$PWD/build/variants/libtrm_prev.so; fi
    run 300 $AB c4 $B: --steps 50 >> $L 2>&1
    run 300 $AB c3x8 $B: --steps 60 --reps 5 >> $L 2>&1
    run 300 $AB c4vg $B: --steps 50 >> $L 2>&1
    run 300 $AB c4vgveg $B: --steps 50 >> $L 2>&1
    TRM_STAGED_SMALL=1 run 300 $AB c3 ${B}_forced: >> $L 2>&1
  done
done
python - <<'PY'
import json
rows = {}
for line in open("gpurun_out/exp25_staged_inputs.log"):
    if line.startswith("{"):
        d = json.loads(line)
        for k, v in d["us_per_step"].items():
            rows.setdefault(d["workload"], {}).setdefault(k, []).append(v["median"])
for wl, r in rows.items():
    for k, v in r.items():
        print(wl, k, v, "mean", round(sum(v) / len(v), 2))
PY
