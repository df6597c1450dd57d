# round 3, GPU call 14: deep columns (two levels per lane) with the late geometry / late boundary-term loads: 76 / 70 VGPRs (6 / 7
# waves per SIMD) instead of 92 / 76 (5 / 6); variant with the 7-wave hint (71 / 70); previous build; deep-column tests first
run() { local limit=$1; shift; timeout -k 10 $limit "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: $*"; exit 1; fi; return 0; }
run 600 python -m pytest tests/test_gpu_deep_columns.py -m gpu -q -x -W ignore::DeprecationWarning > gpurun_out/exp14_tests.log 2>&1; tail -3 gpurun_out/exp14_tests.log
L=gpurun_out/exp14_deep.log; : > $L
for B in new deep7 prev new deep7 prev; do
  case $B in new) unset TRM_LIBRARY;; deep7) export TRM_LIBRARY=$PWD/build/variants/libtrm_deep7.so;; prev) export TRM_LIBRARY=$PWD/build/variants/libtrm_prev.so;; esac
  echo "== $B" >> $L
  run 300 python profiles/tools/deep_timing.py >> $L 2>/dev/null
  echo >> $L
done
python - <<'PY'
import json
name = None
for line in open("gpurun_out/exp14_deep.log"):
    if line.startswith("=="): name = line.split()[1]
    elif line.startswith("{"):
        d = json.loads(line)
        print(name, {k: (v["us_per_step"] if isinstance(v, dict) else v) for k, v in d.items() if "unfused" not in k})
PY
