# round 3, GPU call 12: ballots of single compares combined on the scalar unit, one unordered compare for the NaN check, pinned ring staging for the series row table
# new build vs the previous commit's build (build/variants/libtrm_prev.so), one process per build, alternating, same box
run() { local limit=$1; shift; timeout -k 10 $limit "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: $*"; exit 1; fi; return 0; }
L=gpurun_out/exp12_ab.log; : > $L
run 900 python -m pytest tests -m gpu -q -x -W ignore::DeprecationWarning > gpurun_out/exp12_full.log 2>&1; tail -4 gpurun_out/exp12_full.log
AB="python profiles/tools/ab_options.py"
for wl in c3x8 c5 c3 c4 c2; do
  case $wl in c3x8) S="--steps 60 --reps 5";; c5|c5vg) S="--steps 30 --reps 5";; c4|c4vg) S="--steps 50";; *) S="";; esac
  for B in new prev new prev; do
    if [ $B = new ]; then unset TRM_LIBRARY; else export TRM_LIBRARY=$PWD/build/variants/libtrm_prev.so; fi
    run 300 $AB $wl $B: $S >> $L 2>&1
  done
done
python - <<'PY'
import json
rows = {}
for line in open("gpurun_out/exp12_ab.log"):
    if line.startswith("{"):
        d = json.loads(line)
        for k, v in d["us_per_step"].items():
            rows.setdefault(d["workload"], {}).setdefault(k, []).append(v["median"])
for wl, r in rows.items():
    print(wl, {k: v for k, v in r.items()}, "new/prev", round(min(r["new"]) / min(r["prev"]), 3))
PY
