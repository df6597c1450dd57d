# round 4, experiment 19: the vegetation-coupled LandModel's column launches with the derivation of T / liq (like any other large
# LandModel; round 3's rule had excluded coupled contexts) against TRM_DERIVE_COUPLED=0; one process per sample.  First the coupled tests.
run() { local limit=$1; shift; timeout -k 10 $limit "$@"; local rc=$?; if [ $rc -ne 0 ]; then echo "FAILED ($rc): $*"; tail -30 gpurun_out/r04_exp19_tests.log; exit 1; fi; return 0; }
run 800 python -m pytest tests/test_gpu_coupled_vegetation.py tests/test_gpu_deep_columns.py tests/test_gpu_restart.py tests/test_gpu_series_window.py -m gpu -q -x -W ignore::DeprecationWarning > gpurun_out/r04_exp19_tests.log 2>&1; tail -2 gpurun_out/r04_exp19_tests.log
L=gpurun_out/r04_exp19_derive_coupled.log; : > $L
for round in 1 2 3 4; do
  for K in 0 1; do
    TRM_DERIVE_COUPLED=$K run 300 python profiles/tools/ab_options.py c4vgveg derive_coupled_$K: --steps 50 --reps 7 >> $L 2>&1
  done
done
grep -h "^{" $L | cut -c1-200
