# round 4, experiment 6 (diagnostic): what the wave-uniform RUN-TIME branches on the boundary-condition kinds cost the Euler column
# program.  Throw-away builds with the signature of the bench workloads as compile-time constants (-DTRM_DIAG_BCSIG: 2 = Value on the
# top temperature alone, the C3 physics; 64 = the LandModel wiring, C4) against the shipped library, one process per sample, alternating.
run() { local limit=$1; shift; timeout -k 10 $limit "$@"; local rc=$?; if [ $rc -ne 0 ]; then echo "FAILED ($rc): $*"; exit 1; fi; return 0; }
L=gpurun_out/r04_exp6_bc_signature.log; : > $L
AB="python profiles/tools/ab_options.py"
for round in 1 2 3; do
  for B in shipped sig; do
    if [ $B = shipped ]; then unset TRM_LIBRARY; else export TRM_LIBRARY=$PWD/build/variants/lib_sig_c3.so; fi
    run 300 $AB c3x8 $B: --steps 60 --reps 5 >> $L 2>&1
    run 300 $AB c3 $B: --reps 7 >> $L 2>&1
    if [ $B = sig ]; then export TRM_LIBRARY=$PWD/build/variants/lib_sig_c4.so; fi
    run 300 $AB c4 $B: --steps 50 --reps 7 >> $L 2>&1
  done
done
unset TRM_LIBRARY
grep -h "^{" $L | cut -c1-260
