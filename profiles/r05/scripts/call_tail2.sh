#!/bin/bash
# same-box A/B of the working tree against the previous commit (build/libterrarium_hip_prev.so): in-launch tests first, then one process per sample
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_surface_in_launch.py tests/test_gpu_program_selection.py tests/test_gpu_full_size.py -x -q -m gpu > gpurun_out/tail_tests.log 2>&1
rc=$?
tail -3 gpurun_out/tail_tests.log
[ $rc -ne 0 ] && exit $rc
L=gpurun_out/exp16c_isa_junk_packed.log
: > $L
for rep in 1 2 3; do
  for lib in old new; do
    for spec in "c5 64" "c5 32" "c4 8" "c4 0"; do
      set -- $spec
      if [ $lib = old ]; then export TRM_LIBRARY=$PWD/build/libterrarium_hip_prev.so; else unset TRM_LIBRARY; fi
      extra=""; [ $2 != 0 ] && extra="--shard $2"
      echo "== $lib $1 shard $2 rep $rep" >> $L
      timeout -k 10 300 python profiles/tools/ab_options.py $1 x: --steps 50 --reps 7 $extra >> $L 2>&1 || { tail -5 $L; exit 1; }
    done
  done
done
python3 - $L <<'PY'
import sys, json, collections
res = collections.defaultdict(list)
key = None
for l in open(sys.argv[1]):
    if l.startswith("=="):
        p = l.split(); key = (p[2], p[4], p[1])
    elif l.startswith("{"):
        res[key].append(json.loads(l)["us_per_step"]["x"]["median"])
for (wl, sh) in sorted({(k[0], k[1]) for k in res}):
    o, n = res[(wl, sh, "old")], res[(wl, sh, "new")]
    print(wl, "shard", sh, "old", o, "new", n, "ratio of medians %.3f" % (sorted(n)[len(n)//2] / sorted(o)[len(o)//2]))
PY
