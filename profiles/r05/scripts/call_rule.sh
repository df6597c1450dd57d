#!/bin/bash
# HEAD: surface processes in the launch vs the launch pair at N145, half of it, a third, both hydraulics
set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/exp18_pair_vs_one_launch_head.log
: > $L
for rep in 1 2 3; do
  for wl in c4 c4vg; do
    timeout -k 10 300 python profiles/tools/ab_options.py $wl pair:surface_in_launch=0 one:surface_in_launch=1 --steps 50 --reps 9 >> $L 2>&1 || { tail -5 $L; exit 1; }
    timeout -k 10 300 python profiles/tools/ab_options.py $wl pair:surface_in_launch=0 one:surface_in_launch=1 --steps 50 --reps 9 --shard 2 >> $L 2>&1 || { tail -5 $L; exit 1; }
  done
done
timeout -k 10 300 python profiles/tools/ab_options.py c4 pair:surface_in_launch=0 one:surface_in_launch=1 --steps 50 --reps 9 --shard 3 >> $L 2>&1
timeout -k 10 300 python profiles/tools/ab_options.py c4vg pair:surface_in_launch=0 one:surface_in_launch=1 --steps 50 --reps 9 --shard 3 >> $L 2>&1
grep workload $L | python3 -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l); u = j['us_per_step']
    print(j['workload'], j['columns'], 'pair', u['pair']['median'], 'one', u['one']['median'], 'ratio %.3f' % (u['one']['median'] / u['pair']['median']))
"
