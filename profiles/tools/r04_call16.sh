# round 4, GPU call 16: (a) where the kernel arguments live -- HIP_FORCE_DEV_KERNARG = 0 / 1 / unset on the launch-bound workloads
# (C2: 7 us per step; a 7 119-column C4 shard), one process per sample; (b) the round's profile collection on the current build
COMMIT=$1
run() { local limit=$1; shift; timeout -k 10 $limit "$@"; local rc=$?; if [ $rc -ne 0 ]; then echo "FAILED ($rc): $*"; exit 1; fi; return 0; }
L=gpurun_out/r04_exp11_dev_kernarg.log; : > $L
AB="python profiles/tools/ab_options.py"
for round in 1 2 3; do
  for K in unset 0 1; do
    if [ $K = unset ]; then unset HIP_FORCE_DEV_KERNARG; else export HIP_FORCE_DEV_KERNARG=$K; fi
    run 300 $AB c2 kernarg_$K: --steps 200 --reps 7 >> $L 2>&1
    run 300 $AB c4 kernarg_$K: --steps 100 --reps 7 --shard 8 >> $L 2>&1
    run 300 $AB c3 kernarg_$K: --steps 100 --reps 7 >> $L 2>&1
  done
done
unset HIP_FORCE_DEV_KERNARG
grep -h "^{" $L | cut -c1-220
bash profiles/collect.sh r04 $COMMIT || exit 1
