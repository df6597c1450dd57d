set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout -k 10 600 python -m pytest tests/test_gpu_column_programs.py -x -q > gpurun_out/r05/call20_tests.log 2>&1
rc=$?
tail -5 gpurun_out/r05/call20_tests.log
[ $rc -ne 0 ] && exit $rc
L=gpurun_out/r05/exp9_multistep_signature.log
for wl in c3 c4 c2 c3vg; do
  steps=100; [ $wl = c4 ] && steps=50
  timeout -k 10 300 python profiles/tools/ab_options.py $wl runtime:steps_per_launch=50,bc_signature=0 sig:steps_per_launch=50,bc_signature=1 --steps $steps --reps 7 >> $L 2>&1 || exit 1
done
timeout -k 10 300 python profiles/tools/ab_options.py c4 runtime:steps_per_launch=50,bc_signature=0 sig:steps_per_launch=50,bc_signature=1 --steps 50 --reps 7 --shard 8 >> $L 2>&1 || exit 1
grep -v amdgpu.ids $L
