# round 4, experiment 3: the instruction cuts again, after the fix of the top-array bookkeeping (experiments 1 and 2 ran with the
# LandModel's surface kernel on its gather path: every C4 / C5 number of theirs is void).  Variants: round 3's build; all cuts off;
# all on; all but the flux-condition branches; all but the rare-path branch of the power; neither of the two (no new branch in
# the hot path); that without the scalar lane masks.  Same box, alternating, 3 rounds.
run() { local limit=$1; shift; timeout -k 10 $limit "$@"; local rc=$?; if [ $rc -ne 0 ]; then echo "FAILED ($rc): $*"; exit 1; fi; return 0; }
run 600 python -m pytest tests/test_gpu_column_programs.py tests/test_gpu_single_process.py -m gpu -q -x -W ignore::DeprecationWarning -k "bookkeeping or three_contexts or device_memory" > gpurun_out/r04_exp3_tests.log 2>&1; tail -2 gpurun_out/r04_exp3_tests.log
L=gpurun_out/r04_exp3_cuts.log; : > $L
AB="python profiles/tools/ab_options.py"
for round in 1 2 3; do
  for B in r3 new alloff no_FLUX no_POWRARE nobranch no_MASKS; do
    case $B in r3) export TRM_LIBRARY=$PWD/build/variants/libtrm_r3.so;; new) unset TRM_LIBRARY;; *) export TRM_LIBRARY=$PWD/build/variants/lib_$B.so;; esac
    run 300 $AB c5 $B: --steps 30 --reps 5 >> $L 2>&1
    run 300 $AB c4 $B: --steps 50 --reps 9 >> $L 2>&1
    run 300 $AB c3x8 $B: --steps 60 --reps 7 >> $L 2>&1
    run 300 $AB c3 $B: --reps 9 >> $L 2>&1
  done
done
unset TRM_LIBRARY
python - <<'PY'
import json
rows = {}
for line in open("gpurun_out/r04_exp3_cuts.log"):
    if line.startswith("{"):
        d = json.loads(line)
        for k, v in d["us_per_step"].items():
            rows.setdefault(d["workload"], {}).setdefault(k, []).append(v["median"])
for wl, r in rows.items():
    base = sum(r["r3"]) / len(r["r3"])
    print(wl, " ".join(f"{k}={sum(v)/len(v):.2f}({sum(v)/len(v)/base:.3f})" for k, v in r.items()))
    print("   ", {k: v for k, v in r.items()})
PY
