# round 3, GPU call 22: the packed fp32 step with T AND liq derived (TRM_OPT_DERIVE_CLOSURE_FIELDS = 1: two field reads less) against
# the liquid fraction alone (3, the rule since exp21) and none (0); fp32 tests with option 1 forced through the env-free path first
run() { local limit=$1; shift; timeout -k 10 $limit "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: $*"; exit 1; fi; return 0; }
L=gpurun_out/exp22_derive_both_fp32.log; : > $L
AB="python profiles/tools/ab_options.py"
for round in 1 2 3; do
  run 400 $AB c5 none:derive_closure_fields=0 liq:derive_closure_fields=3 both:derive_closure_fields=1 --steps 30 --reps 5 >> $L 2>&1
  run 400 $AB c5vg none:derive_closure_fields=0 liq:derive_closure_fields=3 both:derive_closure_fields=1 --steps 30 --reps 5 >> $L 2>&1
done
python - <<'PY'
import json
rows = {}
for line in open("gpurun_out/exp22_derive_both_fp32.log"):
    if line.startswith("{"):
        d = json.loads(line)
        for k, v in d["us_per_step"].items():
            rows.setdefault(d["workload"], {}).setdefault(k, []).append(v["median"])
        print(d["workload"], d["status"])
for wl, r in rows.items():
    for k, v in r.items():
        print(wl, k, v, "mean", round(sum(v) / len(v), 1))
PY
