set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout -k 10 600 python -m pytest tests/test_gpu_surface_in_launch.py -x -q > gpurun_out/r05/call4_tests.log 2>&1
rc=$?
tail -15 gpurun_out/r05/call4_tests.log
[ $rc -ne 0 ] && exit $rc
L=gpurun_out/r05/exp3_surface_in_launch.log
for wl in c4 c4vg; do
  timeout -k 10 200 python profiles/tools/ab_options.py $wl pair:surface_in_launch=0 one:surface_in_launch=1 --steps 50 --reps 7 >> $L 2>&1 || exit 1
done
timeout -k 10 200 python profiles/tools/ab_options.py c4 pair:surface_in_launch=0 one:surface_in_launch=1 --steps 50 --reps 7 --shard 8 >> $L 2>&1 || exit 1
grep -v amdgpu.ids $L
