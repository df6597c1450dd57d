#!/bin/bash
# lookahead: tests + same-box A/B on C4 (variants interleaved in one process)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_surface_lookahead.py tests/test_gpu_surface_in_launch.py -x -q -m gpu > gpurun_out/look_tests.log 2>&1
rc=$?
tail -15 gpurun_out/look_tests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python profiles/tools/ab_options.py c4 pair:surface_in_launch=0 one:surface_lookahead=0 look:surface_lookahead=2 --steps 400 --reps 9 > gpurun_out/look_ab.log 2>&1 || { tail -5 gpurun_out/look_ab.log; exit 1; }
cat gpurun_out/look_ab.log
timeout -k 10 300 python profiles/tools/ab_options.py c4 pair:surface_in_launch=0 one:surface_lookahead=0 look:surface_lookahead=2 --steps 400 --reps 9 >> gpurun_out/look_ab.log 2>&1
tail -1 gpurun_out/look_ab.log
