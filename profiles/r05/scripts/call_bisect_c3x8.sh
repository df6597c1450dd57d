#!/bin/bash
# 8 x N145 (HBM-resident C3 physics): round 4's library, two commits of this round and HEAD on one box, one process per sample, five rounds
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=gpurun_out/exp20_c3x8_bisect.log
: > $L
for rep in 1 2 3 4 5; do
  for lib in r04 67b2af3 bd015aa head; do
    if [ $lib = head ]; then unset TRM_LIBRARY; else export TRM_LIBRARY=$GRAFT_REPO_ROOT/build/variants/lib_$lib.so; fi
    echo "== $lib rep $rep" >> $L
    timeout -k 10 300 python profiles/tools/ab_options.py c3x8 x: --steps 50 --reps 5 2>/dev/null | grep workload >> $L || exit 1
  done
done
python3 - $L <<'PY'
import sys, json, collections
res = collections.defaultdict(list); key = None
for l in open(sys.argv[1]):
    if l.startswith("=="): key = l.split()[1]
    elif l.startswith("{"): res[key].append(json.loads(l)["us_per_step"]["x"]["median"])
for k in ("r04", "67b2af3", "bd015aa", "head"): print(k, res[k], "median", sorted(res[k])[len(res[k])//2])
PY
