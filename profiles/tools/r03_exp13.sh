# round 3, GPU call 13: the vegetation-coupled step (c4vgveg) and C2 across three builds on one box -- current, 68cd937 (before the
# scalar-side ballots), 94e99ea (before "no vector load behind the stores") -- one process per build, alternating
run() { local limit=$1; shift; timeout -k 10 $limit "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: $*"; exit 1; fi; return 0; }
L=gpurun_out/exp13_ab.log; : > $L
AB="python profiles/tools/ab_options.py"
for wl in c4vgveg c2 c3vg; do
  for B in new prev old new prev old; do
    case $B in new) unset TRM_LIBRARY;; prev) export TRM_LIBRARY=$PWD/build/variants/libtrm_prev.so;; old) export TRM_LIBRARY=$PWD/build/variants/libtrm_94e99ea.so;; esac
    run 300 $AB $wl $B: --steps 50 >> $L 2>&1
  done
done
python - <<'PY'
import json
rows = {}
for line in open("gpurun_out/exp13_ab.log"):
    if line.startswith("{"):
        d = json.loads(line)
        for k, v in d["us_per_step"].items():
            rows.setdefault(d["workload"], {}).setdefault(k, []).append(v["median"])
for wl, r in rows.items():
    print(wl, r)
PY
