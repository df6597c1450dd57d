set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r05/call8_tests.log 2>&1
rc=$?
tail -8 gpurun_out/r05/call8_tests.log
exit $rc
