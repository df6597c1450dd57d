set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout -k 10 900 python -m pytest tests/test_gpu_surface_in_launch.py tests/test_gpu_program_selection.py -x -q > gpurun_out/r05/call12_tests.log 2>&1
rc=$?
tail -12 gpurun_out/r05/call12_tests.log
[ $rc -ne 0 ] && exit $rc
L=gpurun_out/r05/exp4_packed_surface_in_launch.log
for wl in c5 c5vg; do
  timeout -k 10 300 python profiles/tools/ab_options.py $wl pair:surface_in_launch=0 one:surface_in_launch=1 --steps 50 --reps 5 >> $L 2>&1 || exit 1
done
grep -v amdgpu.ids $L
