set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout -k 10 900 python -m pytest tests/test_gpu_deep_columns.py -x -q > gpurun_out/r05/call16_tests.log 2>&1
rc=$?
tail -12 gpurun_out/r05/call16_tests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python profiles/tools/grouped_timing.py 2 > gpurun_out/r05/deep_columns_timing_hbm.log 2>&1 || { tail -5 gpurun_out/r05/deep_columns_timing_hbm.log; exit 1; }
grep -v "amdgpu.ids\|^{" gpurun_out/r05/deep_columns_timing_hbm.log
