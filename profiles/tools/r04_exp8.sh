# round 4, experiment 8: the per-column inputs requested right behind the field loads, in front of the derivation (TRM_EARLY_INPUTS = 1,
# the shipped build) against round 3's order (build/variants/lib_late.so: -DTRM_EARLY_INPUTS=0, where the boundary values of a wave were
# requested only after it had waited for its fields) and against round 3's library; one process per sample, alternating, three rounds.
# First the tests of the files the change touches.
run() { local limit=$1; shift; timeout -k 10 $limit "$@"; local rc=$?; if [ $rc -ne 0 ]; then echo "FAILED ($rc): $*"; tail -30 gpurun_out/r04_exp8_tests.log; exit 1; fi; return 0; }
run 1000 python -m pytest tests/test_gpu_column_programs.py tests/test_gpu_parity.py -m gpu -q -x -W ignore::DeprecationWarning > gpurun_out/r04_exp8_tests.log 2>&1; tail -3 gpurun_out/r04_exp8_tests.log
L=gpurun_out/r04_exp8_early_inputs.log; : > $L
AB="python profiles/tools/ab_options.py"
for round in 1 2 3; do
  for B in r3 late early; do
    case $B in r3) export TRM_LIBRARY=$PWD/build/variants/libtrm_r3.so;; early) unset TRM_LIBRARY;; *) export TRM_LIBRARY=$PWD/build/variants/lib_$B.so;; esac
    run 300 $AB c3x8 $B: --steps 60 --reps 5 >> $L 2>&1
    run 300 $AB c3 $B: --reps 7 >> $L 2>&1
    run 300 $AB c4 $B: --steps 50 --reps 7 >> $L 2>&1
    run 300 $AB c4vg $B: --steps 50 --reps 7 >> $L 2>&1
    run 300 $AB c5 $B: --steps 30 --reps 5 >> $L 2>&1
  done
done
unset TRM_LIBRARY
python - <<'PY'
import json
rows = {}
for line in open("gpurun_out/r04_exp8_early_inputs.log"):
    if line.startswith("{"):
        d = json.loads(line)
        for k, v in d["us_per_step"].items():
            rows.setdefault(d["workload"], {}).setdefault(k, []).append(v["median"])
for wl, r in rows.items():
    base = sum(r["r3"]) / len(r["r3"])
    print(wl, " ".join(f"{k}={sum(v)/len(v):.2f}({sum(v)/len(v)/base:.3f})" for k, v in r.items()), {k: v for k, v in r.items()})
PY
