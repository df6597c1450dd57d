#!/bin/bash
# granule validity on the scalar unit: tests of the in-launch paths, then pair vs one launch (same box)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_surface_in_launch.py tests/test_gpu_program_selection.py -x -q -m gpu > gpurun_out/valid_tests.log 2>&1
rc=$?
tail -5 gpurun_out/valid_tests.log
[ $rc -ne 0 ] && exit $rc
L=gpurun_out/exp12_scalar_validity.log
: > $L
for rep in 1 2 3; do
  timeout -k 10 300 python profiles/tools/ab_options.py c4 pair:surface_in_launch=0 one:surface_in_launch=1 --steps 50 --reps 9 >> $L 2>&1 || { tail -5 $L; exit 1; }
  timeout -k 10 300 python profiles/tools/ab_options.py c4 pair:surface_in_launch=0 one:surface_in_launch=1 --steps 50 --reps 9 --shard 8 >> $L 2>&1 || { tail -5 $L; exit 1; }
done
timeout -k 10 300 python profiles/tools/ab_options.py c5 pair:surface_in_launch=0 one:surface_in_launch=1 --steps 50 --reps 9 --shard 64 >> $L 2>&1
grep workload $L | python3 -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l); u = j['us_per_step']
    print(j['workload'], j['columns'], 'pair', u['pair']['median'], 'one', u['one']['median'], 'ratio %.3f' % (u['one']['median'] / u['pair']['median']))
"
