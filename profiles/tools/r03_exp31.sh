# round 3, GPU call 31: the per-level geometry records through LDS (one fetch of the table per workgroup) against every lane loading
# its 56 bytes; parity suite of the variant first
run() { local limit=$1; shift; timeout -k 10 $limit "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: $*"; exit 1; fi; return 0; }
TRM_LIBRARY=$PWD/build/variants/libtrm_GEOMLDS.so run 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_column_programs.py tests/test_gpu_full_size.py -m gpu -q -x -W ignore::DeprecationWarning -k "not staged_per_column" > gpurun_out/exp31_tests.log 2>&1; tail -2 gpurun_out/exp31_tests.log
L=gpurun_out/exp31_geometry_lds.log; : > $L
AB="python profiles/tools/ab_options.py"
for round in 1 2 3; do
  for B in plain GEOMLDS; do
    if [ $B = plain ]; then unset TRM_LIBRARY; else export TRM_LIBRARY=$PWD/build/variants/libtrm_$B.so; fi
    run 300 $AB c3 $B: >> $L 2>&1
    run 300 $AB c4 $B: --steps 50 >> $L 2>&1
    run 300 $AB c3x8 $B: --steps 60 --reps 5 >> $L 2>&1
    run 300 $AB c2 $B: >> $L 2>&1
    run 300 $AB c4vgveg $B: --steps 50 >> $L 2>&1
  done
done
python - <<'PY'
import json
rows = {}
for line in open("gpurun_out/exp31_geometry_lds.log"):
    if line.startswith("{"):
        d = json.loads(line)
        for k, v in d["us_per_step"].items():
            rows.setdefault(d["workload"], {}).setdefault(k, []).append(v["median"])
for wl, r in rows.items():
    print(wl, r, "lds/plain", round(sum(r["GEOMLDS"]) / sum(r["plain"]), 3))
PY
