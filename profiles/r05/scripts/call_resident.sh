#!/bin/bash
# resident argument block: a parity subset, then A/B by option on one box (variants interleaved in one process)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_surface_in_launch.py tests/test_gpu_program_selection.py tests/test_gpu_column_programs.py -x -q -m gpu > gpurun_out/resident_tests.log 2>&1
rc=$?
tail -5 gpurun_out/resident_tests.log
[ $rc -ne 0 ] && exit $rc
L=gpurun_out/exp14_resident_arguments.log
: > $L
for rep in 1 2; do
  for wl in c2 c3 c4; do
    timeout -k 10 300 python profiles/tools/ab_options.py $wl byvalue:resident_arguments=0 resident:resident_arguments=1 --steps 50 --reps 9 >> $L 2>&1 || { tail -5 $L; exit 1; }
  done
  timeout -k 10 300 python profiles/tools/ab_options.py c4 byvalue:resident_arguments=0 resident:resident_arguments=1 --steps 50 --reps 9 --shard 8 >> $L 2>&1 || { tail -5 $L; exit 1; }
  timeout -k 10 300 python profiles/tools/ab_options.py c3 byvalue:resident_arguments=0 resident:resident_arguments=1 --steps 50 --reps 9 --shard 8 >> $L 2>&1 || { tail -5 $L; exit 1; }
done
timeout -k 10 300 python profiles/tools/ab_options.py c3x8 byvalue:resident_arguments=0 resident:resident_arguments=1 --steps 50 --reps 5 >> $L 2>&1
grep workload $L | python3 -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l); u = j['us_per_step']
    print(j['workload'], j['columns'], 'byvalue', u['byvalue']['median'], 'resident', u['resident']['median'], 'ratio %.3f' % (u['resident']['median'] / u['byvalue']['median']), 'wall', u['byvalue']['wall_median'], u['resident']['wall_median'])
"
