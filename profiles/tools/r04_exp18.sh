# round 4, experiment 18: k_column_deep with its per-column inputs through the scalar memory path (one column per wave: wave-uniform
# addresses; the shipped build) against vector loads in the middle of the wave's life (build/variants/lib_deepvec.so:
# -DTRM_DEEP_SCALAR_INPUTS=0, round 3's form); one process per sample, alternating.  First the deep-column tests.
run() { local limit=$1; shift; timeout -k 10 $limit "$@"; local rc=$?; if [ $rc -ne 0 ]; then echo "FAILED ($rc): $*"; tail -20 gpurun_out/r04_exp18_tests.log; exit 1; fi; return 0; }
run 800 python -m pytest tests/test_gpu_deep_columns.py tests/test_gpu_coupled_vegetation.py -m gpu -q -x -W ignore::DeprecationWarning > gpurun_out/r04_exp18_tests.log 2>&1; tail -2 gpurun_out/r04_exp18_tests.log
L=gpurun_out/r04_exp18_deep_scalar_inputs.log; : > $L
for round in 1 2 3; do
  for B in vector scalar; do
    case $B in scalar) unset TRM_LIBRARY;; *) export TRM_LIBRARY=$PWD/build/variants/lib_deepvec.so;; esac
    run 400 python profiles/tools/deep_ab.py $B >> $L 2>&1
  done
done
unset TRM_LIBRARY
grep -h "^{" $L
