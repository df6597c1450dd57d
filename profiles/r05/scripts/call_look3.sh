#!/bin/bash
# lookahead workgroups in the middle of the grid: tests, same-box A/B on C4; in-launch vs pair by size for the van Genuchten hydraulics
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_surface_lookahead.py -x -q -m gpu > gpurun_out/look_tests.log 2>&1
rc=$?
tail -5 gpurun_out/look_tests.log
[ $rc -ne 0 ] && exit $rc
L=gpurun_out/exp10b_lookahead_mid.log
: > $L
for rep in 1 2 3; do
  timeout -k 10 300 python profiles/tools/ab_options.py c4 pair:surface_in_launch=0 one:surface_lookahead=0 look:surface_lookahead=2 --steps 50 --reps 9 >> $L 2>&1 || { tail -5 $L; exit 1; }
done
timeout -k 10 300 python profiles/tools/ab_options.py c4vg pair:surface_in_launch=0 one:surface_lookahead=0 look:surface_lookahead=2 --steps 50 --reps 9 >> $L 2>&1 || { tail -5 $L; exit 1; }
for sh in 2 4 8; do
  timeout -k 10 300 python profiles/tools/ab_options.py c4vg pair:surface_in_launch=0 one:surface_in_launch=1,surface_lookahead=0 --steps 50 --reps 9 --shard $sh >> $L 2>&1 || { tail -5 $L; exit 1; }
  timeout -k 10 300 python profiles/tools/ab_options.py c4 pair:surface_in_launch=0 one:surface_in_launch=1,surface_lookahead=0 --steps 50 --reps 9 --shard $sh >> $L 2>&1 || { tail -5 $L; exit 1; }
done
grep workload $L
