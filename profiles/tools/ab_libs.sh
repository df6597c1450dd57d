#!/bin/bash
# Same-box A/B of library BUILDS: one process per sample (a process loads one library), the order of the builds drawn afresh for every
# workload and round -- the second of two HBM-resident runs in a row reads ~8 % slower than a run that follows light work
# (EXPERIMENTS R5.9), so a fixed "old, then new" order is biased against the new build.
#   bash profiles/tools/ab_libs.sh LOG ROUNDS "name=path/to/lib.so name2=HEAD ..." "wl[:shard] wl ..."      (HEAD: the shipped library)
# Prints, per workload, every build's samples and the ratio of medians against the first build.
set -o pipefail
LOG=$1; ROUNDS=$2; LIBS=$3; WLS=$4
: > $LOG
for rep in $(seq 1 $ROUNDS); do
  for spec in $WLS; do
    wl=${spec%%:*}; shard=""; [[ $spec == *:* ]] && shard="--shard ${spec##*:}"
    steps=50; [ $wl = c3 ] && steps=100; [ $wl = c2 ] && steps=100
    for lib in $(echo $LIBS | tr ' ' '\n' | shuf); do
      name=${lib%%=*}; path=${lib##*=}
      if [ "$path" = HEAD ]; then unset TRM_LIBRARY; else export TRM_LIBRARY=$PWD/$path; fi
      echo "== $name $spec rep $rep" >> $LOG
      timeout -k 10 300 python profiles/tools/ab_options.py $wl x: --steps $steps --reps 5 $shard 2>/dev/null | grep workload >> $LOG || exit 1
    done
  done
done
python3 - $LOG "$LIBS" <<'PY'
import sys, json, collections
res = collections.defaultdict(list); key = None
for l in open(sys.argv[1]):
    if l.startswith("=="):
        p = l.split(); key = (p[2], p[1])
    elif l.startswith("{"):
        res[key].append(json.loads(l)["us_per_step"]["x"]["median"])
names = [x.split("=")[0] for x in sys.argv[2].split()]
med = lambda v: sorted(v)[len(v) // 2]
for wl in sorted({k[0] for k in res}):
    print(wl, " | ".join(f"{n} {res[(wl, n)]}" + ("" if n == names[0] else " ratio %.3f" % (med(res[(wl, n)]) / med(res[(wl, names[0])]))) for n in names))
PY
