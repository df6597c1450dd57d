# round 4, experiment 20: the cell phase of k_surface_veg (plant available water of every cell) with its loads and stores out of the
# per-pass branches (where conditional stores met the compiler had put a full vmcnt wait: every pass waited for the previous
# pass's store) against the previous commit's build (build/variants/lib_base.so); one process per sample.  First the vegetation tests.
run() { local limit=$1; shift; timeout -k 10 $limit "$@"; local rc=$?; if [ $rc -ne 0 ]; then echo "FAILED ($rc): $*"; tail -30 gpurun_out/r04_exp20_tests.log; exit 1; fi; return 0; }
run 800 python -m pytest tests/test_gpu_coupled_vegetation.py tests/test_gpu_vegetation.py tests/test_gpu_deep_columns.py tests/test_gpu_restart.py -m gpu -q -x -W ignore::DeprecationWarning > gpurun_out/r04_exp20_tests.log 2>&1; tail -2 gpurun_out/r04_exp20_tests.log
L=gpurun_out/r04_exp20_paw_cell_phase.log; : > $L
for round in 1 2 3 4; do
  for B in base new; do
    case $B in new) unset TRM_LIBRARY;; *) export TRM_LIBRARY=$PWD/build/variants/lib_$B.so;; esac
    run 300 python profiles/tools/ab_options.py c4vgveg $B: --steps 50 --reps 7 >> $L 2>&1
  done
done
unset TRM_LIBRARY
grep -h "^{" $L | cut -c1-200
