# round 4, experiment 13: small grids (one wave per SIMD, latency-bound) with their per-column inputs on the VECTOR path, requested with the
# fields, instead of scalar loads requested behind them (TRM_SMALL_GRID_VECTOR_INPUTS = column count up to which); one process per sample
run() { local limit=$1; shift; timeout -k 10 $limit "$@"; local rc=$?; if [ $rc -ne 0 ]; then echo "FAILED ($rc): $*"; exit 1; fi; return 0; }
L=gpurun_out/r04_exp13_small_grid_vector_inputs.log; : > $L
AB="python profiles/tools/ab_options.py"
for round in 1 2 3; do
  for K in 0 100000; do
    export TRM_SMALL_GRID_VECTOR_INPUTS=$K
    run 300 $AB c2 vector_upto_$K: --steps 200 --reps 7 >> $L 2>&1
    run 300 $AB c4 vector_upto_$K: --steps 100 --reps 7 --shard 8 >> $L 2>&1
    run 300 $AB c4vgveg vector_upto_$K: --steps 50 --reps 7 >> $L 2>&1
    run 300 $AB c3 vector_upto_$K: --steps 100 --reps 7 --shard 8 >> $L 2>&1
  done
done
grep -h "^{" $L | cut -c1-200
