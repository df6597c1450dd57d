# round 3, GPU call 26: the per-column inputs of the column program through the scalar memory path (s_load per column of the wave)
# against the previous build (vector loads, every lane of a column the same address); full suite first
run() { local limit=$1; shift; timeout -k 10 $limit "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: $*"; exit 1; fi; return 0; }
run 900 python -m pytest tests -m gpu -q -x -W ignore::DeprecationWarning > gpurun_out/exp26_tests.log 2>&1; tail -2 gpurun_out/exp26_tests.log
L=gpurun_out/exp26_scalar_inputs.log; : > $L
AB="python profiles/tools/ab_options.py"
for round in 1 2 3; do
  for B in new prev; do
    if [ $B = new ]; then unset TRM_LIBRARY; else export TRM_LIBRARY=$PWD/build/variants/libtrm_prev.so; fi
    run 300 $AB c4 $B: --steps 50 >> $L 2>&1
    run 300 $AB c3x8 $B: --steps 60 --reps 5 >> $L 2>&1
    run 300 $AB c3 $B: >> $L 2>&1
    run 300 $AB c4vg $B: --steps 50 >> $L 2>&1
    run 300 $AB c2 $B: >> $L 2>&1
    run 300 $AB c4vgveg $B: --steps 50 >> $L 2>&1
  done
done
python - <<'PY'
import json
rows = {}
for line in open("gpurun_out/exp26_scalar_inputs.log"):
    if line.startswith("{"):
        d = json.loads(line)
        for k, v in d["us_per_step"].items():
            rows.setdefault(d["workload"], {}).setdefault(k, []).append(v["median"])
for wl, r in rows.items():
    print(wl, r, "new/prev", round(sum(r["new"]) / sum(r["prev"]), 3))
PY
