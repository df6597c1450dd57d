#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/gpu_suite.log 2>&1
rc=$?
tail -3 gpurun_out/gpu_suite.log
[ $rc -ne 0 ] && exit $rc
bash profiles/collect.sh r05 $(cat profiles/r05/scripts/HEAD_COMMIT)
