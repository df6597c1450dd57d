# round 3, GPU call 16: XCD-aware workgroup -> column mapping (every XCD one contiguous eighth of the columns, TRM_XCD_REMAP = 1)
# against the plain mapping; one process per sample, alternating; parity suite of the variant first
run() { local limit=$1; shift; timeout -k 10 $limit "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: $*"; exit 1; fi; return 0; }
TRM_LIBRARY=$PWD/build/variants/libtrm_xcd.so run 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py -m gpu -q -x -W ignore::DeprecationWarning > gpurun_out/exp16_tests.log 2>&1; tail -2 gpurun_out/exp16_tests.log
L=gpurun_out/exp16_xcd.log; : > $L
AB="python profiles/tools/ab_options.py"
for round in 1 2 3; do
  for B in plain xcd; do
    if [ $B = plain ]; then unset TRM_LIBRARY; else export TRM_LIBRARY=$PWD/build/variants/libtrm_xcd.so; fi
    run 300 $AB c3x8 $B: --steps 60 --reps 5 >> $L 2>&1
    run 300 $AB c5 $B: --steps 30 --reps 5 >> $L 2>&1
    run 300 $AB c3 $B: >> $L 2>&1
    run 300 $AB c4 $B: --steps 50 >> $L 2>&1
  done
done
python - <<'PY'
import json
rows = {}
for line in open("gpurun_out/exp16_xcd.log"):
    if line.startswith("{"):
        d = json.loads(line)
        for k, v in d["us_per_step"].items():
            rows.setdefault(d["workload"], {}).setdefault(k, []).append(v["median"])
for wl, r in rows.items():
    for k, v in r.items():
        print(wl, k, v, "mean", round(sum(v) / len(v), 2))
PY
