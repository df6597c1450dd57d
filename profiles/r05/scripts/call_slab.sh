#!/bin/bash
# per-process modes of the C3 step: twelve processes with the fields allocated one by one (shipped) and twelve with ONE slab, alternating
set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/exp17_field_slab.log
: > $L
for i in $(seq 1 12); do
  for slab in 0 1; do
    echo "== slab $slab process $i" >> $L
    TRM_FIELD_SLAB=$slab timeout -k 10 200 python profiles/tools/ab_options.py c3 x: --steps 100 --reps 7 >> $L 2>&1 || { tail -5 $L; exit 1; }
  done
done
for i in $(seq 1 4); do
  for slab in 0 1; do
    echo "== slab $slab c2 process $i" >> $L
    TRM_FIELD_SLAB=$slab timeout -k 10 200 python profiles/tools/ab_options.py c2 x: --steps 100 --reps 7 >> $L 2>&1 || { tail -5 $L; exit 1; }
  done
done
python3 - $L <<'PY'
import sys, json, collections
res = collections.defaultdict(list); key = None
for l in open(sys.argv[1]):
    if l.startswith("=="):
        p = l.split(); key = (p[2], "c2" if "c2" in l else "c3")
    elif l.startswith("{"):
        res[key].append(json.loads(l)["us_per_step"]["x"]["median"])
for k in sorted(res): print(k, sorted(res[k]))
PY
