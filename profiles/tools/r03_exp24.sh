# round 3, GPU call 24 (diagnostic build, wrong results by construction): the column program without its per-column 8-byte LOADS
# (boundary values, ground heat flux, infiltration, surface excess water, skin temperature replaced by constants)
run() { local limit=$1; shift; timeout -k 10 $limit "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: $*"; exit 1; fi; return 0; }
L=gpurun_out/exp24_diag_small_loads.log; : > $L
AB="python profiles/tools/ab_options.py"
for round in 1 2 3; do
  for B in full NOLD; do
    if [ $B = full ]; then unset TRM_LIBRARY; else export TRM_LIBRARY=$PWD/build/variants/libtrm_$B.so; fi
    run 300 $AB c4 $B: --steps 30 >> $L 2>&1
    run 300 $AB c3 $B: --steps 30 >> $L 2>&1
    run 300 $AB c3x8 $B: --steps 30 --reps 5 >> $L 2>&1
  done
done
python - <<'PY'
import json
rows = {}
for line in open("gpurun_out/exp24_diag_small_loads.log"):
    if line.startswith("{"):
        d = json.loads(line)
        for k, v in d["us_per_step"].items():
            rows.setdefault(d["workload"], {}).setdefault(k, []).append(v["median"])
for wl, r in rows.items():
    for k, v in r.items():
        print(wl, k, v, "mean", round(sum(v) / len(v), 2))
PY
