# round 4, experiment 12: (a) the level records requested BEHIND the field loads (in the instances that read T / liq the record came
# first and a write-after-write hazard on its unused eighth value made the wave wait for it before it requested its fields);
# (b) the boundary-condition signature compiled into the instances that read T / liq as well (small grids, the vegetation-coupled
# LandModel), whose per-column scalar loads each sat in a branch with a wait of its own.  Against the previous commit's build
# (build/variants/lib_base.so), one process per sample, alternating, three rounds.  First the tests of the files the change touches.
run() { local limit=$1; shift; timeout -k 10 $limit "$@"; local rc=$?; if [ $rc -ne 0 ]; then echo "FAILED ($rc): $*"; tail -30 gpurun_out/r04_exp12_tests.log; exit 1; fi; return 0; }
run 1000 python -m pytest tests/test_gpu_column_programs.py tests/test_gpu_parity.py tests/test_gpu_coupled_vegetation.py -m gpu -q -x -W ignore::DeprecationWarning > gpurun_out/r04_exp12_tests.log 2>&1; tail -3 gpurun_out/r04_exp12_tests.log
L=gpurun_out/r04_exp12_records_behind_fields.log; : > $L
AB="python profiles/tools/ab_options.py"
for round in 1 2 3; do
  for B in base new; do
    case $B in new) unset TRM_LIBRARY;; *) export TRM_LIBRARY=$PWD/build/variants/lib_$B.so;; esac
    run 300 $AB c2 $B: --steps 200 --reps 7 >> $L 2>&1
    run 300 $AB c4vgveg $B: --steps 50 --reps 7 >> $L 2>&1
    run 300 $AB c4 $B: --steps 100 --reps 7 --shard 8 >> $L 2>&1
    run 300 $AB c3 $B: --reps 7 >> $L 2>&1
    run 300 $AB c4 $B: --steps 50 --reps 7 >> $L 2>&1
    run 300 $AB c3x8 $B: --steps 60 --reps 5 >> $L 2>&1
  done
done
unset TRM_LIBRARY
python - <<'PY'
import json
rows = {}
for line in open("gpurun_out/r04_exp12_records_behind_fields.log"):
    if line.startswith("{"):
        d = json.loads(line)
        for k, v in d["us_per_step"].items():
            rows.setdefault((d["workload"], d["columns"]), {}).setdefault(k, []).append(v["median"])
for wl, r in rows.items():
    base = sum(r["base"]) / len(r["base"])
    print(wl, " ".join(f"{k}={sum(v)/len(v):.2f}({sum(v)/len(v)/base:.3f})" for k, v in r.items()), {k: v for k, v in r.items()})
PY
