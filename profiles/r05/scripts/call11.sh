set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout -k 10 900 python -m pytest tests/test_gpu_state_functions.py tests/test_gpu_restart.py -x -q > gpurun_out/r05/call11_tests.log 2>&1
rc=$?
tail -30 gpurun_out/r05/call11_tests.log
exit $rc
