# round 4, experiment 5: the fp64 column program with the pressure head derived as well (TRM_OPT_DERIVE_CLOSURE_FIELDS = 5: two field
# reads per cell instead of three; the compile-time r^(-5) of round 3 makes the formula far cheaper than when r2 measured this)
# against the library's rule (2: T and liq derived), alternating in one process
run() { local limit=$1; shift; timeout -k 10 $limit "$@"; local rc=$?; if [ $rc -ne 0 ]; then echo "FAILED ($rc): $*"; exit 1; fi; return 0; }
run 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "pressure_head_derived" -W ignore::DeprecationWarning > gpurun_out/r04_exp5_tests.log 2>&1; tail -3 gpurun_out/r04_exp5_tests.log
L=gpurun_out/r04_exp5_derive_psi_fp64.log; : > $L
AB="python profiles/tools/ab_options.py"
for round in 1 2 3; do
  run 300 $AB c3x8 t_liq:derive_closure_fields=2 all:derive_closure_fields=5 --steps 60 --reps 5 >> $L 2>&1
  run 300 $AB c3 t_liq:derive_closure_fields=2 all:derive_closure_fields=5 --steps 100 --reps 7 >> $L 2>&1
  run 300 $AB c4 t_liq:derive_closure_fields=2 all:derive_closure_fields=5 --steps 50 --reps 7 >> $L 2>&1
  run 300 $AB c3vg t_liq:derive_closure_fields=2 all:derive_closure_fields=5 --steps 100 --reps 7 >> $L 2>&1
done
grep -h "^{" $L | cut -c1-300
