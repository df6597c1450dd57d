# round 3, GPU call 15b: field skew, more samples at 8 x N145 (one process per sample, alternating)
run() { local limit=$1; shift; timeout -k 10 $limit "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: $*"; exit 1; fi; return 0; }
L=gpurun_out/exp15b_skew.log; : > $L
AB="python profiles/tools/ab_options.py"
for round in 1 2 3 4 5; do
  for K in 0 8448 16640 33024 16384; do
    TRM_FIELD_SKEW=$K run 300 $AB c3x8 skew$K: --steps 60 --reps 5 >> $L 2>&1
  done
done
python - <<'PY'
import json
rows = {}
for line in open("gpurun_out/exp15b_skew.log"):
    if line.startswith("{"):
        d = json.loads(line)
        for k, v in d["us_per_step"].items():
            rows.setdefault(d["workload"], {}).setdefault(k, []).append(v["median"])
for wl, r in rows.items():
    for k, v in r.items():
        print(wl, k, v, "mean", round(sum(v) / len(v), 1))
PY
