# round 4, GPU call 4: deep-column coverage (Heun with generic boundary kinds, the vegetation-coupled LandModel), the series-window
# declaration; then the A/B of call 3 (round 3's build / the new kernels / the SGPR-capped variant) and the instruction counts
run() { local limit=$1; shift; timeout -k 10 $limit "$@"; local rc=$?; if [ $rc -ne 0 ]; then echo "FAILED ($rc): $*"; exit 1; fi; return 0; }
run 900 python -m pytest tests/test_gpu_deep_columns.py tests/test_gpu_series_window.py tests/test_gpu_coupled_vegetation.py tests/test_gpu_restart.py -m gpu -q -x -W ignore::DeprecationWarning > gpurun_out/r04_call4_tests.log 2>&1; tail -3 gpurun_out/r04_call4_tests.log
L=gpurun_out/r04_exp1_valu_cut.log; : > $L
AB="python profiles/tools/ab_options.py"
for round in 1 2 3; do
  for B in r3 new sg90; do
    case $B in r3) export TRM_LIBRARY=$PWD/build/variants/libtrm_r3.so;; new) unset TRM_LIBRARY;; sg90) export TRM_LIBRARY=$PWD/build/variants/lib_sg90.so;; esac
    run 300 $AB c3 $B: >> $L 2>&1
    run 300 $AB c3x8 $B: --steps 60 --reps 5 >> $L 2>&1
    run 300 $AB c4 $B: --steps 50 >> $L 2>&1
    run 300 $AB c2 $B: >> $L 2>&1
    run 300 $AB c3vg $B: >> $L 2>&1
    run 300 $AB c5 $B: --steps 30 --reps 5 >> $L 2>&1
  done
done
unset TRM_LIBRARY
python - <<'PY'
import json
rows = {}
for line in open("gpurun_out/r04_exp1_valu_cut.log"):
    if line.startswith("{"):
        d = json.loads(line)
        for k, v in d["us_per_step"].items():
            rows.setdefault(d["workload"], {}).setdefault(k, []).append(v["median"])
for wl, r in rows.items():
    print(wl, {k: v for k, v in r.items()}, "new/r3", round(sum(r["new"]) / sum(r["r3"]), 3), "sg90/r3", round(sum(r["sg90"]) / sum(r["r3"]), 3))
PY
for wl in c3 c3x8 c4 c5; do
  bash profiles/tools/pmc_count.sh r3 $wl $PWD/build/variants/libtrm_r3.so || exit 1
  bash profiles/tools/pmc_count.sh new $wl || exit 1
done
