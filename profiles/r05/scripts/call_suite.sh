#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/gpu_suite.log 2>&1
rc=$?
tail -8 gpurun_out/gpu_suite.log
exit $rc
