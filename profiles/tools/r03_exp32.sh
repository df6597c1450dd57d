# round 3, GPU call 32: 512-thread workgroups (16 columns: the staged per-column outputs become full 128-byte lines) against 256
run() { local limit=$1; shift; timeout -k 10 $limit "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: $*"; exit 1; fi; return 0; }
TRM_LIBRARY=$PWD/build/variants/libtrm_B512.so run 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_column_programs.py -m gpu -q -x -W ignore::DeprecationWarning -k "not staged_per_column" > gpurun_out/exp32_tests.log 2>&1; tail -2 gpurun_out/exp32_tests.log
L=gpurun_out/exp32_block512.log; : > $L
AB="python profiles/tools/ab_options.py"
for round in 1 2 3; do
  for B in b256 B512; do
    if [ $B = b256 ]; then unset TRM_LIBRARY; else export TRM_LIBRARY=$PWD/build/variants/libtrm_$B.so; fi
    run 300 $AB c3x8 $B: --steps 60 --reps 5 >> $L 2>&1
    run 300 $AB c4 $B: --steps 50 >> $L 2>&1
    run 300 $AB c3 $B: >> $L 2>&1
    run 300 $AB c5 $B: --steps 30 --reps 5 >> $L 2>&1
  done
done
python - <<'PY'
import json
rows = {}
for line in open("gpurun_out/exp32_block512.log"):
    if line.startswith("{"):
        d = json.loads(line)
        for k, v in d["us_per_step"].items():
            rows.setdefault(d["workload"], {}).setdefault(k, []).append(v["median"])
for wl, r in rows.items():
    print(wl, r, "512/256", round(sum(r["B512"]) / sum(r["b256"]), 3))
PY
