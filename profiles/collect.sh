#!/bin/bash
# Collects the round's profile on the GPU box (run through gpurun from the repo root):
#   bash profiles/collect.sh r01
# kernel trace + stats, then the PMC counters in separate passes (never combined with a trace), then the
# summary JSON the bench's roofline.traffic is read from.  Output: gpurun_out/prof_<round>/ (scratch);
# copy kernel_stats / pmc_summary / bench_*.json from there into profiles/<round>/.
set -e
ROUND=${1:-r01}
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$ROUND
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python bench.py --steps 100 --warmup 10 > $OUT/bench_c3.json 2> $OUT/bench_c3.err
for wl in c2 c3vg c4 c4vg c5 c5vg; do python bench.py --workload $wl --no-cpu-baseline --steps 50 --warmup 5 > $OUT/bench_$wl.json 2> $OUT/bench_$wl.err; done
python bench.py --kernel unfused --no-cpu-baseline > $OUT/bench_c3_unfused.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python bench.py --no-cpu-baseline --steps 100 --warmup 10 > $OUT/bench_trace.json 2> $OUT/trace.err
B="python bench.py --no-cpu-baseline --steps 20 --warmup 2"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $B > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $B > /dev/null 2> $OUT/pmc_write.err
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH --output-format csv -d $OUT/pmc_sq1 -- $B > /dev/null 2> $OUT/pmc_sq1.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq2 -- $B > /dev/null 2> $OUT/pmc_sq2.err
python profiles/summarize_pmc.py $OUT > $OUT/pmc_summary_c3_fused.json
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_c3_fused.csv \;
echo done; cut -c1-400 $OUT/bench_c3.json
