# round 4, GPU call 5: the whole suite on the shipped defaults (instruction cuts without the two branch-introducing ones), then
# (a) the packed fp32 step with the pressure head derived as well (TRM_OPT_DERIVE_CLOSURE_FIELDS = 4) against the liquid fraction alone
#     (2 = the library's rule at C5) and none, C5 and C5-VG, alternating in one process;
# (b) the coupling exchange at BASELINE config 4's shard size (profiles/tools/coupling_exchange.py)
run() { local limit=$1; shift; timeout -k 10 $limit "$@"; local rc=$?; if [ $rc -ne 0 ]; then echo "FAILED ($rc): $*"; exit 1; fi; return 0; }
run 900 python -m pytest tests -m gpu -q -x -W ignore::DeprecationWarning > gpurun_out/r04_call5_tests.log 2>&1; tail -3 gpurun_out/r04_call5_tests.log
L=gpurun_out/r04_exp4_derive_psi_fp32.log; : > $L
AB="python profiles/tools/ab_options.py"
for round in 1 2 3; do
  run 300 $AB c5 none:derive_closure_fields=0 liq:derive_closure_fields=3 liq_psi:derive_closure_fields=4 --steps 30 --reps 5 >> $L 2>&1
  run 300 $AB c5vg none:derive_closure_fields=0 liq:derive_closure_fields=3 liq_psi:derive_closure_fields=4 --steps 30 --reps 5 >> $L 2>&1
done
grep -h "^{" $L | cut -c1-400
run 600 python profiles/tools/coupling_exchange.py > gpurun_out/r04_coupling_exchange.log 2>&1; tail -2 gpurun_out/r04_coupling_exchange.log | cut -c1-1500
