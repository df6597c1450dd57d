#!/bin/bash
# instruction counts and wave cycles of the vegetation-coupled 0-D kernel (separate --pmc passes, no tracing):
#   bash profiles/tools/pmc_veg.sh      (through gpurun from the repo root)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_veg
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_LDS --output-format csv -d $OUT/a -- python3 profiles/tools/veg_timing.py 20 > /dev/null 2> $OUT/a.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/b -- python3 profiles/tools/veg_timing.py 20 > /dev/null 2> $OUT/b.err
python3 - <<PY
import csv, glob, collections
for sub in ("a", "b"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
    for path in glob.glob("$OUT/%s/**/*counter_collection.csv" % sub, recursive=True):
        for row in csv.DictReader(open(path)):
            k = row["Kernel_Name"].split("(")[0][:60]
            if "k_surface" in k or "k_plant" in k:
                acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); n[k].add(row["Dispatch_Id"])
    for k, c in acc.items():
        d = len(n[k])
        print(sub, k, "dispatches", d, " per dispatch:", " ".join(f"{x} {c[x]/d:.0f}" for x in sorted(c)))
PY
