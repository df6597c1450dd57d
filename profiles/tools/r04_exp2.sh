# round 4, experiment 2: which of the eight instruction cuts of experiment 1 costs the HBM-resident and LandModel steps their time?
# (exp 1: -27 ... -40 vector instructions per wave, C3 / C3-VG / C2 2 % faster, but 8 x N145 +5.5 %, C4 +9 %, C5 +19 %)
# Leave-one-out builds (every cut on but one), all cuts off, round 3's build; same box, alternating.
run() { local limit=$1; shift; timeout -k 10 $limit "$@"; local rc=$?; if [ $rc -ne 0 ]; then echo "FAILED ($rc): $*"; exit 1; fi; return 0; }
L=gpurun_out/r04_exp2_leave_one_out.log; : > $L
AB="python profiles/tools/ab_options.py"
for round in 1 2; do
  for B in r3 new alloff no_MASKS no_WT no_NSZ no_FLUX no_CHECK no_POWRARE no_FRAC no_SOFF; do
    case $B in r3) export TRM_LIBRARY=$PWD/build/variants/libtrm_r3.so;; new) unset TRM_LIBRARY;; *) export TRM_LIBRARY=$PWD/build/variants/lib_$B.so;; esac
    run 300 $AB c5 $B: --steps 30 --reps 5 >> $L 2>&1
    run 300 $AB c4 $B: --steps 50 >> $L 2>&1
    run 300 $AB c3x8 $B: --steps 60 --reps 5 >> $L 2>&1
    run 300 $AB c3 $B: >> $L 2>&1
  done
done
unset TRM_LIBRARY
python - <<'PY'
import json
rows = {}
for line in open("gpurun_out/r04_exp2_leave_one_out.log"):
    if line.startswith("{"):
        d = json.loads(line)
        for k, v in d["us_per_step"].items():
            rows.setdefault(d["workload"], {}).setdefault(k, []).append(v["median"])
for wl, r in rows.items():
    base = sum(r["r3"]) / len(r["r3"])
    print(wl, " ".join(f"{k}={sum(v)/len(v):.1f}({sum(v)/len(v)/base:.3f})" for k, v in r.items()))
PY
