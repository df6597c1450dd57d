# round 3, GPU call 11: SGPR cap of the step kernels (amdgpu_num_sgpr 96 / 80 / 64) -- the SPI counters say the dispatcher is
# blocked by SGPR allocation (SPI_RA_SGPR_SIMD_FULL_CSN) and the mean occupancy is 5.5 of 8 waves per SIMD
run() { local limit=$1; shift; timeout -k 10 $limit "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: $*"; exit 1; fi; return 0; }
L=gpurun_out/exp11_ab.log; : > $L
AB="python profiles/tools/ab_options.py"
for wl in c3x8 c5 c3 c4; do
  case $wl in c3x8) S="--steps 60 --reps 5";; c5) S="--steps 30 --reps 5";; c4) S="--steps 50";; *) S="";; esac
  for B in 96 80 64 96 80 64; do
    export TRM_LIBRARY=$PWD/build/variants/libtrm_sgpr$B.so
    run 300 $AB $wl s$B: $S >> $L 2>&1
  done
done
python - <<'PY'
import json
rows = {}
for line in open("gpurun_out/exp11_ab.log"):
    if line.startswith("{"):
        d = json.loads(line)
        for k, v in d["us_per_step"].items():
            rows.setdefault(d["workload"], {}).setdefault(k, []).append(v["median"])
for wl, r in rows.items():
    print(wl, r)
PY
cd /tmp && export TMPDIR=/tmp
for B in 96 80 64; do
  export TRM_LIBRARY=$GRAFT_REPO_ROOT/build/variants/libtrm_sgpr$B.so
  for wl in c3x8 c5; do
    rm -rf /tmp/occ; timeout -k 10 120 rocprofv3 --pmc MeanOccupancyPerCU --output-format csv -d /tmp/occ -- python $GRAFT_REPO_ROOT/bench.py --workload $wl --no-cpu-baseline --no-hbm-resident --multistep 0 --steps 10 --warmup 2 --spinup-ms 0 --repeats 1 > /dev/null 2>&1
    python - $B $wl <<'PY'
import csv, glob, sys
vals = {}
for p in glob.glob("/tmp/occ/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if "trm::k_" in r["Kernel_Name"]:
            vals.setdefault(r["Kernel_Name"].split("(")[0][:60], []).append(float(r["Counter_Value"]))
for k, v in vals.items():
    if len(v) > 5: print("cap", sys.argv[1], sys.argv[2], k, "MeanOccupancyPerCU", round(sum(v) / len(v), 2), len(v))
PY
  done
done
