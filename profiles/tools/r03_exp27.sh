# round 3, GPU call 27: the packed fp32 step with its per-column inputs through the scalar memory path (TRM_SCALAR_INPUTS_PK = 1)
# against two vector loads per input; fp32 land tests under the switch first
run() { local limit=$1; shift; timeout -k 10 $limit "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: $*"; exit 1; fi; return 0; }
TRM_SCALAR_INPUTS_PK=1 run 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_column_programs.py tests/test_gpu_full_size.py -m gpu -q -x -W ignore::DeprecationWarning -k "not staged_per_column" > gpurun_out/exp27_tests.log 2>&1; tail -2 gpurun_out/exp27_tests.log
L=gpurun_out/exp27_scalar_inputs_pk.log; : > $L
AB="python profiles/tools/ab_options.py"
for round in 1 2 3; do
  for T in 0 1; do
    TRM_SCALAR_INPUTS_PK=$T run 300 $AB c5 s$T: --steps 30 --reps 5 >> $L 2>&1
    TRM_SCALAR_INPUTS_PK=$T run 300 $AB c5vg s$T: --steps 30 --reps 5 >> $L 2>&1
    TRM_SCALAR_INPUTS_PK=$T run 300 $AB c5 s$T: --steps 30 --reps 5 --shard 64 >> $L 2>&1
  done
done
python - <<'PY'
import json
rows = {}
for line in open("gpurun_out/exp27_scalar_inputs_pk.log"):
    if line.startswith("{"):
        d = json.loads(line)
        for k, v in d["us_per_step"].items():
            rows.setdefault((d["workload"], d["columns"]), {}).setdefault(k, []).append(v["median"])
for wl, r in rows.items():
    for k, v in r.items():
        print(wl, k, v, "mean", round(sum(v) / len(v), 2))
PY
