# per-wave instruction counts (PMC) of k_step_wave for every library under build/variants
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for f in build/variants/*.so; do
  OUT=$GRAFT_REPO_ROOT/gpurun_out/cnt_$(basename $f .so); rm -rf $OUT; mkdir -p $OUT
  export TRM_LIBRARY=$GRAFT_REPO_ROOT/$f
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_ACTIVE_INST_ANY --output-format csv -d $OUT -- python bench.py --no-cpu-baseline --steps 20 --warmup 2 "$@" > /dev/null 2> $OUT/err.txt
  python - "$OUT" "$f" <<'PY'
import csv, glob, os, sys
out, name = sys.argv[1], sys.argv[2]
sums = {}
for path in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        if "k_step_wave" in r["Kernel_Name"]:
            sums[r["Counter_Name"]] = sums.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
w = sums.get("SQ_WAVES", 1.0)
print(name, " ".join(f"{k[3:].replace('INSTS_','').lower()}={v / w:.1f}" for k, v in sorted(sums.items()) if k != "SQ_WAVES"))
PY
done
