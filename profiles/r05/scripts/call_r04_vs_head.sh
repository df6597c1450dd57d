#!/bin/bash
# round 4's library (commit ae4e7e0, rebuilt: build/variants/lib_r04.so) against HEAD's on one box: one process per sample, alternating
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=gpurun_out/exp19_round4_library_vs_head_final.log
: > $L
for rep in 1 2 3; do
  for spec in "c4 0" "c4vg 0" "c4 8" "c3 0" "c3x8 0" "c5 0" "c2 0"; do
    set -- $spec
    steps=50; [ $1 = c3 ] && steps=100; [ $1 = c2 ] && steps=100
    extra=""; [ $2 != 0 ] && extra="--shard $2"
    for lib in r04 head; do
      if [ $lib = head ]; then unset TRM_LIBRARY; else export TRM_LIBRARY=$GRAFT_REPO_ROOT/build/variants/lib_r04.so; fi
      echo "== $lib $1 shard $2 rep $rep" >> $L
      timeout -k 10 300 python profiles/tools/ab_options.py $1 x: --steps $steps --reps 5 $extra 2>/dev/null | grep workload >> $L || exit 1
    done
  done
done
python3 - $L <<'PY'
import sys, json, collections
res = collections.defaultdict(list); key = None
for l in open(sys.argv[1]):
    if l.startswith("=="):
        p = l.split(); key = (p[2], p[4], p[1])
    elif l.startswith("{"):
        res[key].append(json.loads(l)["us_per_step"]["x"]["median"])
for (wl, sh) in sorted({(k[0], k[1]) for k in res}):
    o, n = res[(wl, sh, "r04")], res[(wl, sh, "head")]
    print(wl, "shard", sh, "r04", o, "head", n, "ratio of medians %.3f" % (sorted(n)[len(n)//2] / sorted(o)[len(o)//2]))
PY
