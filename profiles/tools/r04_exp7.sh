# round 4, experiment 7: the boundary-condition signature compiled into the per-step programs (TRM_OPT_BC_SIGNATURE, BCSIG) against the
# same library reading the kinds at run time (option 0), alternating in one process; first the tests of the new instances.
run() { local limit=$1; shift; timeout -k 10 $limit "$@"; local rc=$?; if [ $rc -ne 0 ]; then echo "FAILED ($rc): $*"; tail -30 gpurun_out/r04_exp7_tests.log; exit 1; fi; return 0; }
run 900 python -m pytest tests/test_gpu_column_programs.py tests/test_host_and_abi.py -m gpu -q -x -W ignore::DeprecationWarning > gpurun_out/r04_exp7_tests.log 2>&1; tail -3 gpurun_out/r04_exp7_tests.log
L=gpurun_out/r04_exp7_bc_signature.log; : > $L
AB="python profiles/tools/ab_options.py"
for round in 1 2 3; do
  run 300 $AB c3x8 runtime:bc_signature=0 compiled:bc_signature=1 --steps 60 --reps 5 >> $L 2>&1
  run 300 $AB c3 runtime:bc_signature=0 compiled:bc_signature=1 --steps 100 --reps 7 >> $L 2>&1
  run 300 $AB c4 runtime:bc_signature=0 compiled:bc_signature=1 --steps 50 --reps 7 >> $L 2>&1
  run 300 $AB c4vg runtime:bc_signature=0 compiled:bc_signature=1 --steps 50 --reps 7 >> $L 2>&1
  run 300 $AB c5 runtime:bc_signature=0 compiled:bc_signature=1 --steps 30 --reps 5 >> $L 2>&1
  run 300 $AB c3vg runtime:bc_signature=0 compiled:bc_signature=1 --steps 100 --reps 7 >> $L 2>&1
done
grep -h "^{" $L | cut -c1-300
