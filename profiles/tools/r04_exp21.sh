# round 4, experiment 21: k_surface with every per-column access in the scalar-base + 32-bit-offset form (22 of 24; 14 vector
# instructions less) against the previous commit's build; one process per sample.  First the LandModel tests.
run() { local limit=$1; shift; timeout -k 10 $limit "$@"; local rc=$?; if [ $rc -ne 0 ]; then echo "FAILED ($rc): $*"; tail -30 gpurun_out/r04_exp21_tests.log; exit 1; fi; return 0; }
run 900 python -m pytest tests/test_gpu_reference_tests.py tests/test_gpu_parity.py tests/test_gpu_column_programs.py tests/test_gpu_full_size.py -m gpu -q -x -W ignore::DeprecationWarning > gpurun_out/r04_exp21_tests.log 2>&1; tail -2 gpurun_out/r04_exp21_tests.log
L=gpurun_out/r04_exp21_surface_saddr.log; : > $L
for round in 1 2 3 4; do
  for B in base new; do
    case $B in new) unset TRM_LIBRARY;; *) export TRM_LIBRARY=$PWD/build/variants/lib_$B.so;; esac
    run 300 python profiles/tools/ab_options.py c4 $B: --steps 50 --reps 7 >> $L 2>&1
    run 300 python profiles/tools/ab_options.py c5 $B: --steps 30 --reps 5 >> $L 2>&1
  done
done
unset TRM_LIBRARY
python - <<'PY'
import json
rows = {}
for line in open("gpurun_out/r04_exp21_surface_saddr.log"):
    if line.startswith("{"):
        d = json.loads(line)
        for k, v in d["us_per_step"].items():
            rows.setdefault((d["workload"], d["columns"]), {}).setdefault(k, []).append(v["median"])
for wl, r in rows.items():
    base = sum(r["base"]) / len(r["base"])
    print(wl, " ".join(f"{k}={sum(v)/len(v):.2f}({sum(v)/len(v)/base:.3f})" for k, v in r.items()), {k: v for k, v in r.items()})
PY
