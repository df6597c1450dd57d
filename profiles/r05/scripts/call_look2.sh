#!/bin/bash
# lookahead: same-box A/B on C4 (variants interleaved in one process), default and van Genuchten hydraulics
set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/exp10_lookahead.log
: > $L
for rep in 1 2 3; do
  timeout -k 10 300 python profiles/tools/ab_options.py c4 pair:surface_in_launch=0 one:surface_lookahead=0 look:surface_lookahead=2 --steps 50 --reps 9 >> $L 2>&1 || { tail -5 $L; exit 1; }
done
timeout -k 10 300 python profiles/tools/ab_options.py c4vg pair:surface_in_launch=0 one:surface_lookahead=0 look:surface_lookahead=2 --steps 50 --reps 9 >> $L 2>&1 || { tail -5 $L; exit 1; }
grep workload $L
