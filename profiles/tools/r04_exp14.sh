# round 4, experiment 14: the packed fp32 step compiled without machine sinking (its eight per-column scalar loads issued as one batch
# with one wait, instead of three groups each next to its use) against the previous commit's build (build/variants/lib_base.so);
# one process per sample, alternating, five rounds (the HBM-resident workloads are bimodal per process: placement).  First its tests.
run() { local limit=$1; shift; timeout -k 10 $limit "$@"; local rc=$?; if [ $rc -ne 0 ]; then echo "FAILED ($rc): $*"; tail -30 gpurun_out/r04_exp14_tests.log; exit 1; fi; return 0; }
run 1000 python -m pytest tests/test_gpu_column_programs.py tests/test_gpu_parity.py tests/test_gpu_full_size.py -m gpu -q -x -W ignore::DeprecationWarning -k "fp32 or packed or f32 or float32 or c5 or C5" > gpurun_out/r04_exp14_tests.log 2>&1; tail -3 gpurun_out/r04_exp14_tests.log
L=gpurun_out/r04_exp14_packed_no_machine_sink.log; : > $L
AB="python profiles/tools/ab_options.py"
for round in 1 2 3 4 5; do
  for B in base new; do
    case $B in new) unset TRM_LIBRARY;; *) export TRM_LIBRARY=$PWD/build/variants/lib_$B.so;; esac
    run 300 $AB c5 $B: --steps 30 --reps 5 >> $L 2>&1
    run 300 $AB c5vg $B: --steps 30 --reps 5 >> $L 2>&1
  done
done
unset TRM_LIBRARY
python - <<'PY'
import json
rows = {}
for line in open("gpurun_out/r04_exp14_packed_no_machine_sink.log"):
    if line.startswith("{"):
        d = json.loads(line)
        for k, v in d["us_per_step"].items():
            rows.setdefault((d["workload"], d["columns"]), {}).setdefault(k, []).append(v["median"])
for wl, r in rows.items():
    base = sum(r["base"]) / len(r["base"])
    print(wl, " ".join(f"{k}={sum(v)/len(v):.2f}({sum(v)/len(v)/base:.3f})" for k, v in r.items()), {k: v for k, v in r.items()})
PY
