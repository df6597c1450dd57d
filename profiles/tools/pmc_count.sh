#!/bin/bash
# Instruction counts per wave of the step kernels of one library on one workload (one --pmc pass, no tracing):
#   bash profiles/tools/pmc_count.sh LABEL WORKLOAD [library.so]     -> gpurun_out/pmc_count_LABEL_WORKLOAD.txt
LABEL=$1; WL=$2; LIB=$3
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmcc_${LABEL}_$WL
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
if [ -n "$LIB" ]; then export TRM_LIBRARY=$LIB; else unset TRM_LIBRARY; fi
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_LDS --output-format csv -d $OUT -- python bench.py --workload $WL --no-cpu-baseline --no-hbm-resident --multistep 0 --steps 20 --warmup 2 --spinup-ms 0 --repeats 1 > /dev/null 2> $OUT.err
rc=$?; if [ $rc -ne 0 ]; then echo "FAILED ($rc): pmc_count $LABEL $WL"; tail -5 $OUT.err; exit 1; fi
python - "$OUT" "$LABEL" "$WL" <<'PY' | tee gpurun_out/pmc_count_${LABEL}_$WL.txt
import csv, glob, collections, sys, os
out, label, wl = sys.argv[1:4]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for path in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(path)):
        k = row["Kernel_Name"].split("(")[0][:90]
        if "k_column" in k or "k_step" in k or "k_surface" in k or "k_land" in k:
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); n[k].add(row["Dispatch_Id"])
for k, c in sorted(acc.items(), key=lambda kv: -len(n[kv[0]])):
    w = c["SQ_WAVES"] or 1
    print(label, wl, k, "dispatches", len(n[k]), "per wave:", " ".join(f"{x[9:].lower()} {c[x]/w:.1f}" for x in sorted(c) if x.startswith("SQ_INSTS")))
PY
rm -rf $OUT
