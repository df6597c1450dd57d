#!/bin/bash
# Collects the round's profile on the GPU box (run through gpurun from the repo root):
#   bash profiles/collect.sh r02 <commit>
# kernel trace + stats, then the PMC counters in separate passes (never combined with a trace), then the summary JSONs
# the bench's roofline.traffic fields are read from.  Output: gpurun_out/prof_<round>/ (scratch); copy kernel_stats_* /
# pmc_summary_* / bench_*.json from there into profiles/<round>/.
set -e
ROUND=${1:-r02}
COMMIT=${2:-unknown}
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$ROUND
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python bench.py --steps 100 --warmup 10 > $OUT/bench_c3.json 2> $OUT/bench_c3.err
for wl in c2 c3vg c4 c4vg c5vg c4vgveg; do python bench.py --workload $wl --no-cpu-baseline --no-hbm-resident --multistep 0 --steps 50 --warmup 5 > $OUT/bench_$wl.json 2> $OUT/bench_$wl.err; done
python bench.py --integrator heun --no-cpu-baseline --no-hbm-resident --multistep 0 > $OUT/bench_c3_heun.json 2>/dev/null
python bench.py --kernel unfused --no-cpu-baseline --no-hbm-resident --multistep 0 > $OUT/bench_c3_unfused.json 2>/dev/null
echo benches done
# kernel trace of the default bench command (C3 headline + its HBM-resident companions c3x8 and c5 in the same run)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python bench.py --no-cpu-baseline --multistep 0 --steps 100 --warmup 10 > $OUT/bench_trace.json 2> $OUT/trace.err
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_default_bench.csv \;
echo trace done
# counters: one workload per command so that dispatch counts stay small; FETCH and WRITE in separate passes
for wl in c3 c3x8 c5 c4vg c5vg; do
  B="python bench.py --workload $wl --no-cpu-baseline --no-hbm-resident --multistep 0 --steps 20 --warmup 2 --spinup-ms 0"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_${wl}/fetch -- $B > /dev/null 2> $OUT/pmc_${wl}_fetch.err
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_${wl}/write -- $B > /dev/null 2> $OUT/pmc_${wl}_write.err
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH --output-format csv -d $OUT/pmc_${wl}/sq1 -- $B > /dev/null 2> $OUT/pmc_${wl}_sq1.err
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_${wl}/sq2 -- $B > /dev/null 2> $OUT/pmc_${wl}_sq2.err
  python profiles/summarize_pmc.py $OUT/pmc_${wl} $wl $COMMIT > $OUT/pmc_summary_${wl}_fused.json
  echo pmc $wl done
done
cut -c1-600 $OUT/bench_c3.json
