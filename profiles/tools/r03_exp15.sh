# round 3, GPU call 15: field base addresses skewed against each other (TRM_FIELD_SKEW bytes x field id) -- do the nine streams of
# the step collide on HBM channels / banks when every field starts on the same alignment?  One process per skew, alternating.
run() { local limit=$1; shift; timeout -k 10 $limit "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: $*"; exit 1; fi; return 0; }
L=gpurun_out/exp15_skew.log; : > $L
AB="python profiles/tools/ab_options.py"
for wl in c3x8 c5; do
  case $wl in c3x8) S="--steps 60 --reps 5";; c5) S="--steps 30 --reps 5";; esac
  for round in 1 2; do
    for K in 0 256 768 4352 16640 66304 1048832; do
      TRM_FIELD_SKEW=$K run 300 $AB $wl skew$K: $S >> $L 2>&1
    done
  done
done
python - <<'PY'
import json
rows = {}
for line in open("gpurun_out/exp15_skew.log"):
    if line.startswith("{"):
        d = json.loads(line)
        for k, v in d["us_per_step"].items():
            rows.setdefault(d["workload"], {}).setdefault(k, []).append(v["median"])
for wl, r in rows.items():
    print(wl, r)
PY
