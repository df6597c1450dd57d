#!/bin/bash
# quick instruction-count comparison of two step-kernel variants on one workload:
#   bash profiles/tools/pmc_quick.sh c3vg
# (separate --pmc pass, no tracing; run through gpurun from the repo root)
WL=${1:-c3}
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmcq_$WL
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for variant in column legacy; do
  TRM_AB_VARIANT=$variant rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_LDS --output-format csv -d $OUT/$variant -- python profiles/tools/one_variant.py $WL > /dev/null 2> $OUT/$variant.err
done
python - <<PY
import csv, glob, collections
for variant in ("column", "legacy"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
    for path in glob.glob("$OUT/%s/**/*counter_collection.csv" % variant, recursive=True):
        for row in csv.DictReader(open(path)):
            k = row["Kernel_Name"].split("(")[0][:70]
            if "k_column" in k or "k_step" in k or "k_surface" in k:
                acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); n[k].add(row["Dispatch_Id"])
    for k, c in acc.items():
        w = c["SQ_WAVES"] or 1
        print(variant, k, "dispatches", len(n[k]), " per wave:", " ".join(f"{x[9:].lower()} {c[x]/w:.1f}" for x in sorted(c) if x.startswith("SQ_INSTS")))
PY
