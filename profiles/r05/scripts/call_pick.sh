#!/bin/bash
# the column pick under an execution mask: parity subset, then previous commit against the working tree, order drawn per round
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_surface_in_launch.py tests/test_gpu_column_programs.py tests/test_gpu_full_size.py tests/test_gpu_reference_tests.py -x -q -m gpu > gpurun_out/pick_tests.log 2>&1
rc=$?
tail -3 gpurun_out/pick_tests.log
[ $rc -ne 0 ] && exit $rc
bash profiles/tools/ab_libs.sh gpurun_out/exp24_flux_selects.log 5 "prev=build/variants/lib_prev.so new=HEAD" "c4 c4vg c4:8"
