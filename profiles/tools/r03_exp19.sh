# round 3, GPU call 19 (diagnostic builds, wrong results by construction): the seven per-column 8-byte stores of the column program as
# ONE 64-byte record per column (four 16-byte stores by the top lane) against the shipped build and against no small stores at all
run() { local limit=$1; shift; timeout -k 10 $limit "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: $*"; exit 1; fi; return 0; }
L=gpurun_out/exp19_aos_record.log; : > $L
AB="python profiles/tools/ab_options.py"
for round in 1 2 3; do
  for B in full AOS NO_BOTH; do
    if [ $B = full ]; then unset TRM_LIBRARY; else export TRM_LIBRARY=$PWD/build/variants/libtrm_$B.so; fi
    run 300 $AB c4 $B: --steps 30 >> $L 2>&1
    run 300 $AB c3 $B: --steps 30 >> $L 2>&1
    run 300 $AB c3x8 $B: --steps 30 --reps 5 >> $L 2>&1
  done
done
python - <<'PY'
import json
rows = {}
for line in open("gpurun_out/exp19_aos_record.log"):
    if line.startswith("{"):
        d = json.loads(line)
        for k, v in d["us_per_step"].items():
            rows.setdefault(d["workload"], {}).setdefault(k, []).append(v["median"])
for wl, r in rows.items():
    for k, v in r.items():
        print(wl, k, v, "mean", round(sum(v) / len(v), 2))
PY
