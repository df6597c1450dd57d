# round 3, GPU call 15c: field skew scan (multiples of 256 B, odd channel shifts) at 8 x N145, and C3 / C5 / C4 at the best ones
run() { local limit=$1; shift; timeout -k 10 $limit "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: $*"; exit 1; fi; return 0; }
L=gpurun_out/exp15c_skew.log; : > $L
AB="python profiles/tools/ab_options.py"
for round in 1 2 3; do
  for K in 0 2304 4352 8448 12544 16640 24832 49408; do
    TRM_FIELD_SKEW=$K run 300 $AB c3x8 skew$K: --steps 60 --reps 5 >> $L 2>&1
  done
done
for round in 1 2 3; do
  for K in 0 8448 16640; do
    TRM_FIELD_SKEW=$K run 300 $AB c3 skew$K: >> $L 2>&1
    TRM_FIELD_SKEW=$K run 300 $AB c5 skew$K: --steps 30 --reps 5 >> $L 2>&1
    TRM_FIELD_SKEW=$K run 300 $AB c4 skew$K: --steps 50 >> $L 2>&1
  done
done
python - <<'PY'
import json
rows = {}
for line in open("gpurun_out/exp15c_skew.log"):
    if line.startswith("{"):
        d = json.loads(line)
        for k, v in d["us_per_step"].items():
            rows.setdefault(d["workload"], {}).setdefault(k, []).append(v["median"])
for wl, r in rows.items():
    for k, v in r.items():
        print(wl, k, v, "mean", round(sum(v) / len(v), 2))
PY
