# Diagnostic counter passes for the dominant step kernel of a workload (wave allocation stalls, wave / VMEM levels, L1-L2 request
# latencies, L2-fabric stalls):   bash profiles/tools/pmc_diag.sh c3x8 [c5 ...]      -> gpurun_out/pmc_diag_<wl>.json
run() { local limit=$1; shift; timeout -k 10 $limit "$@"; local rc=$?; if [ $rc -ne 0 ]; then echo "FAILED ($rc): $*"; exit 1; fi; return 0; }
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc_diag; rm -rf $OUT; mkdir -p $OUT; cd $R
for wl in "$@"; do
  B="python bench.py --workload $wl --no-cpu-baseline --no-hbm-resident --multistep 0 --steps 12 --warmup 2 --spinup-ms 0 --repeats 1"
  n=0
  SETS=${PMC_SETS:-"SPI_RA_WAVE_SIMD_FULL_CSN SPI_RA_VGPR_SIMD_FULL_CSN;SPI_RA_REQ_NO_ALLOC_CSN SPI_RA_RES_STALL_CSN;SQ_LEVEL_WAVES SQ_INST_LEVEL_VMEM SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES;TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum;TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum;TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum;TCC_BUSY_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum;MeanOccupancyPerCU;MemUnitStalled VALUBusy"}
  IFS=';' read -ra SETLIST <<< "$SETS"
  for set in "${SETLIST[@]}"; do
    n=$((n+1))
    run 120 rocprofv3 --pmc $set --output-format csv -d $OUT/$wl/p$n -- $B > /dev/null 2> $OUT/${wl}_p$n.err
  done
  python - "$OUT/$wl" "$wl" > $R/gpurun_out/pmc_diag_$wl.json <<'PY'
import csv, glob, json, os, sys
d, wl = sys.argv[1], sys.argv[2]
rows = []
for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    rows += list(csv.DictReader(open(path)))
freq = {}
for r in rows:
    if "trm::k_" in r["Kernel_Name"]:
        freq[r["Kernel_Name"]] = freq.get(r["Kernel_Name"], 0) + 1
dom = max(freq, key=freq.get)
acc = {}
for r in rows:
    if r["Kernel_Name"] == dom:
        acc.setdefault(r["Counter_Name"], {}).setdefault(r["Dispatch_Id"], 0.0)
        acc[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
print(json.dumps({"workload": wl, "kernel": dom.split("(")[0], "per_dispatch_mean": {k: sum(v.values()) / len(v) for k, v in sorted(acc.items())}}, indent=1))
PY
  cat $R/gpurun_out/pmc_diag_$wl.json
done
rm -rf $OUT
