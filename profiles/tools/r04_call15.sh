# round 4, GPU call 15: the whole GPU suite on the current build, then the packed fp32 step with its field pointers fetched in one batch
# (all eight field loads back to back) against the build of two commits ago (build/variants/lib_prev.so), C5 and C5-VG
run() { local limit=$1; shift; timeout -k 10 $limit "$@"; local rc=$?; if [ $rc -ne 0 ]; then echo "FAILED ($rc): $*"; tail -40 gpurun_out/r04_call15_tests.log; exit 1; fi; return 0; }
run 1100 python -m pytest tests -m gpu -q -x -W ignore::DeprecationWarning > gpurun_out/r04_call15_tests.log 2>&1; tail -3 gpurun_out/r04_call15_tests.log
L=gpurun_out/r04_exp10_packed_pointers.log; : > $L
AB="python profiles/tools/ab_options.py"
for round in 1 2 3; do
  for B in prev new; do
    case $B in new) unset TRM_LIBRARY;; *) export TRM_LIBRARY=$PWD/build/variants/lib_$B.so;; esac
    run 300 $AB c5 $B: --steps 30 --reps 5 >> $L 2>&1
    run 300 $AB c5vg $B: --steps 30 --reps 5 >> $L 2>&1
  done
done
unset TRM_LIBRARY
grep -h "^{" $L | cut -c1-230
