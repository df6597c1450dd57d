# round 3, GPU call 21: the packed fp32 step with the incoming liquid fraction derived (TRM_OPT_DERIVE_CLOSURE_FIELDS = 3) against
# read, re-measured on the final kernels (store ordering, field skew); two contexts of each per process, three processes
run() { local limit=$1; shift; timeout -k 10 $limit "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: $*"; exit 1; fi; return 0; }
L=gpurun_out/exp21_derive_liq_fp32.log; : > $L
AB="python profiles/tools/ab_options.py"
for round in 1 2 3; do
  run 400 $AB c5 none1:derive_closure_fields=0 liq1:derive_closure_fields=3 none2:derive_closure_fields=0 liq2:derive_closure_fields=3 --steps 30 --reps 5 >> $L 2>&1
  run 400 $AB c5vg none1:derive_closure_fields=0 liq1:derive_closure_fields=3 none2:derive_closure_fields=0 liq2:derive_closure_fields=3 --steps 30 --reps 5 >> $L 2>&1
done
python - <<'PY'
import json
rows = {}
for line in open("gpurun_out/exp21_derive_liq_fp32.log"):
    if line.startswith("{"):
        d = json.loads(line)
        for k, v in d["us_per_step"].items():
            rows.setdefault(d["workload"], {}).setdefault(k[:-1], []).append(v["median"])
for wl, r in rows.items():
    for k, v in r.items():
        print(wl, k, v, "mean", round(sum(v) / len(v), 1))
PY
