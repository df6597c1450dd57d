#!/bin/bash
# packed fp32 step: masks handed to the repair -- fp32 tests, then the c5 / c5vg legs of the collection again (the kernels changed)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_column_programs.py tests/test_gpu_surface_in_launch.py tests/test_gpu_full_size.py tests/test_gpu_parity.py tests/test_gpu_program_selection.py -x -q -m gpu > gpurun_out/pk_tests.log 2>&1
rc=$?
tail -3 gpurun_out/pk_tests.log
[ $rc -ne 0 ] && exit $rc
WLS="c5 c5vg" bash profiles/collect.sh r05 $(cat profiles/r05/scripts/HEAD_COMMIT)
