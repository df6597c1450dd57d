# round 4, experiment 17: every line of the kernel argument segment requested at the top of the fp64 column programs (dummy scalar
# loads, one wait: build/variants/lib_kprefetch.so, -DTRM_KERNARG_PREFETCH=1) against the shipped library; one process per sample
run() { local limit=$1; shift; timeout -k 10 $limit "$@"; local rc=$?; if [ $rc -ne 0 ]; then echo "FAILED ($rc): $*"; exit 1; fi; return 0; }
L=gpurun_out/r04_exp17_kernarg_prefetch.log; : > $L
AB="python profiles/tools/ab_options.py"
for round in 1 2 3 4; do
  for B in shipped kprefetch; do
    case $B in shipped) unset TRM_LIBRARY;; *) export TRM_LIBRARY=$PWD/build/variants/lib_$B.so;; esac
    run 300 $AB c2 $B: --steps 200 --reps 7 >> $L 2>&1
    run 300 $AB c3 $B: --reps 7 >> $L 2>&1
    run 300 $AB c4 $B: --steps 50 --reps 7 >> $L 2>&1
    run 300 $AB c4 $B: --steps 100 --reps 7 --shard 8 >> $L 2>&1
    run 300 $AB c3x8 $B: --steps 60 --reps 5 >> $L 2>&1
  done
done
unset TRM_LIBRARY
python - <<'PY'
import json
rows = {}
for line in open("gpurun_out/r04_exp17_kernarg_prefetch.log"):
    if line.startswith("{"):
        d = json.loads(line)
        for k, v in d["us_per_step"].items():
            rows.setdefault((d["workload"], d["columns"]), {}).setdefault(k, []).append(v["median"])
for wl, r in rows.items():
    base = sum(r["shipped"]) / len(r["shipped"])
    print(wl, " ".join(f"{k}={sum(v)/len(v):.2f}({sum(v)/len(v)/base:.3f})" for k, v in r.items()), {k: v for k, v in r.items()})
PY
