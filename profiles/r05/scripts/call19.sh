set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout -k 10 600 python -m pytest tests/test_gpu_surface_in_launch.py -x -q -k "uneven or under_load" > gpurun_out/r05/call19_tests.log 2>&1
rc=$?
tail -12 gpurun_out/r05/call19_tests.log
exit $rc
