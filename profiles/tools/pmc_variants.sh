#!/bin/bash
# VALU / wave of the Euler column program with T / liq read and derived (separate --pmc pass, no tracing):
#   bash profiles/tools/pmc_variants.sh c3
WL=${1:-c3}
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmcv_$WL
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for variant in column derive multi50; do
  TRM_AB_VARIANT=$variant rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT/$variant -- python profiles/tools/one_variant.py $WL > /dev/null 2> $OUT/$variant.err
done
python - <<PY
import csv, glob, collections
for tag in ("column", "derive", "multi50"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for path in glob.glob("$OUT/%s/**/*counter_collection.csv" % tag, recursive=True):
        for row in csv.DictReader(open(path)):
            k = row["Kernel_Name"].split("(")[0][:70]
            if "k_column" in k:
                acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
    for k, c in acc.items():
        w = c["SQ_WAVES"] or 1
        print(tag, k, " per wave:", " ".join(f"{x[9:].lower()} {c[x]/w:.1f}" for x in sorted(c) if x.startswith("SQ_INSTS")))
PY
