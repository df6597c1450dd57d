# round 3, GPU call 28: final build (staged outputs and input path compile-time) against 9e9dea7, same box; full suite first
run() { local limit=$1; shift; timeout -k 10 $limit "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: $*"; exit 1; fi; return 0; }
L=gpurun_out/exp29_compile_time_choice.log; : > $L
run 900 python -m pytest tests -m gpu -q -x -W ignore::DeprecationWarning > gpurun_out/exp29_tests.log 2>&1; tail -2 gpurun_out/exp29_tests.log
AB="python profiles/tools/ab_options.py"
for round in 1 2 3; do
  for B in new prev; do
    if [ $B = new ]; then unset TRM_LIBRARY; else export TRM_LIBRARY=$PWD/build/variants/libtrm_prev.so; fi
    run 300 $AB c3x8 $B: --steps 60 --reps 5 >> $L 2>&1
    run 300 $AB c5 $B: --steps 30 --reps 5 >> $L 2>&1
    run 300 $AB c5vg $B: --steps 30 --reps 5 >> $L 2>&1
    run 300 $AB c4 $B: --steps 50 >> $L 2>&1
    run 300 $AB c3 $B: >> $L 2>&1
  done
done
python - <<'PY'
import json
rows = {}
for line in open("gpurun_out/exp29_compile_time_choice.log"):
    if line.startswith("{"):
        d = json.loads(line)
        for k, v in d["us_per_step"].items():
            rows.setdefault(d["workload"], {}).setdefault(k, []).append(v["median"])
for wl, r in rows.items():
    print(wl, r, "new/prev", round(sum(r["new"]) / sum(r["prev"]), 3))
PY
