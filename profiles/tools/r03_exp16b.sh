# round 3, GPU call 16b: the packed fp32 step with (default build) and without (libtrm_nopk.so) the XCD-aware mapping: C5, C5-VG,
# a 7 119-column fp32 shard; fp32 tests of the default build first
run() { local limit=$1; shift; timeout -k 10 $limit "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: $*"; exit 1; fi; return 0; }
run 900 python -m pytest tests -m gpu -q -x -W ignore::DeprecationWarning > gpurun_out/exp16b_tests.log 2>&1; tail -2 gpurun_out/exp16b_tests.log
L=gpurun_out/exp16b_xcd.log; : > $L
AB="python profiles/tools/ab_options.py"
for round in 1 2 3; do
  for B in remap plain; do
    if [ $B = remap ]; then unset TRM_LIBRARY; else export TRM_LIBRARY=$PWD/build/variants/libtrm_nopk.so; fi
    run 300 $AB c5 $B: --steps 30 --reps 5 >> $L 2>&1
    run 300 $AB c5vg $B: --steps 30 --reps 5 >> $L 2>&1
  done
done
python - <<'PY'
import json
rows = {}
for line in open("gpurun_out/exp16b_xcd.log"):
    if line.startswith("{"):
        d = json.loads(line)
        for k, v in d["us_per_step"].items():
            rows.setdefault(d["workload"], {}).setdefault(k, []).append(v["median"])
for wl, r in rows.items():
    for k, v in r.items():
        print(wl, k, v, "mean", round(sum(v) / len(v), 2))
PY
