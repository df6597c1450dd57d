#!/bin/bash
# the LandModel step at the end of the round: round 4's library against HEAD's (order drawn per round), then HEAD's launch pair against its one launch
set -o pipefail
mkdir -p gpurun_out
bash profiles/tools/ab_libs.sh gpurun_out/exp25_round4_vs_head_c4_final.log 6 "r04=build/variants/lib_r04.so head=HEAD" "c4 c4vg c4:8" || exit 1
L=gpurun_out/exp25b_pair_vs_one_launch_final.log
: > $L
for rep in 1 2 3; do
  for wl in c4 c4vg; do
    timeout -k 10 300 python profiles/tools/ab_options.py $wl pair:surface_in_launch=0 one:surface_in_launch=1 --steps 50 --reps 9 >> $L 2>&1 || exit 1
  done
done
grep workload $L | python3 -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l); u = j['us_per_step']
    print(j['workload'], j['columns'], 'pair', u['pair']['median'], 'one', u['one']['median'], 'ratio %.3f' % (u['one']['median'] / u['pair']['median']))
"
