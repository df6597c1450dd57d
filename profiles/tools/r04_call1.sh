# round 4, GPU call 1: the library split into 17 translation units -- whole GPU suite, then same-box baseline numbers of the
# per-step kernels (the split build; ISA identical to round 3's)
run() { local limit=$1; shift; timeout -k 10 $limit "$@"; local rc=$?; if [ $rc -ne 0 ]; then echo "FAILED ($rc): $*"; exit 1; fi; return 0; }
run 900 python -m pytest tests -m gpu -q -x -W ignore::DeprecationWarning > gpurun_out/r04_call1_tests.log 2>&1; tail -3 gpurun_out/r04_call1_tests.log
L=gpurun_out/r04_call1_baseline.log; : > $L
AB="python profiles/tools/ab_options.py"
for round in 1 2; do
  run 300 $AB c3 base: >> $L 2>&1
  run 300 $AB c3x8 base: --steps 60 --reps 5 >> $L 2>&1
  run 300 $AB c4 base: --steps 50 >> $L 2>&1
  run 300 $AB c5 base: --steps 30 --reps 5 >> $L 2>&1
done
grep -h "^{" $L | cut -c1-220
