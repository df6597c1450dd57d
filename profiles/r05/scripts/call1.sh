set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout -k 10 600 python -m pytest tests/test_gpu_tail_surface.py -x -q > gpurun_out/r05/call1_tests.log 2>&1
rc=$?
tail -15 gpurun_out/r05/call1_tests.log
[ $rc -ne 0 ] && exit $rc
for wl in c4 c4vg; do
  timeout -k 10 200 python profiles/tools/ab_options.py $wl pair:tail_surface=0 tail:tail_surface=1 --steps 50 --reps 7 >> gpurun_out/r05/exp1_tail_surface.log 2>&1 || exit 1
done
timeout -k 10 200 python profiles/tools/ab_options.py c4 pair:tail_surface=0 tail:tail_surface=1 --steps 50 --reps 7 --shard 8 >> gpurun_out/r05/exp1_tail_surface.log 2>&1 || exit 1
cat gpurun_out/r05/exp1_tail_surface.log
