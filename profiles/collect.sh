#!/bin/bash
# Collects the round's profile on the GPU box (run through gpurun from the repo root):
#   bash profiles/collect.sh r04 <commit>
# Per workload, in this order:
#   1. PMC counters in separate rocprofv3 --pmc passes (never combined with a trace) -> pmc_summary_<wl>_fused.json, copied
#      into profiles/<round>/ ON THE BOX so that the bench runs below read the traffic of THIS kernel build;
#   2. ONE `rocprofv3 --kernel-trace --stats` run of `bench.py --workload <wl> --no-hbm-resident --no-single-process --multistep 0
#      --no-cpu-baseline` -> kernel_stats_<wl>.csv: the dominant kernel's average is one row, one launch size;
#   3. the bench line of the same command without the profiler -> bench_<wl>.json.
# Output: gpurun_out/prof_<round>/ (scratch); copy kernel_stats_* / pmc_summary_* / bench_*.json into profiles/<round>/.
ROUND=${1:-r05}
COMMIT=${2:-unknown}
# A step that fails for ANY reason (a timeout, an abort or a fault under the profiler: rc 134 / 139, a Python error) ends the collection:
# no GPU work follows a faulted step, and no summary is made of partial counters.
run() { local limit=$1; shift; timeout -k 10 $limit "$@"; local rc=$?; if [ $rc -ne 0 ]; then echo "FAILED ($rc): $*"; ls -t $OUT/*.err 2>/dev/null | head -1 | xargs -r tail -5; exit 1; fi; return 0; }
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$ROUND
[ -z "$WLS" ] && rm -rf $OUT; mkdir -p $OUT $GRAFT_REPO_ROOT/profiles/$ROUND
cd $GRAFT_REPO_ROOT
steps_of() { case $1 in c4|c4vg|c4vgveg|c5|c5vg) echo 50;; c3x8) echo 60;; *) echo 100;; esac; }
for wl in ${WLS:-c3 c3x8 c5 c4 c4vg c5vg c2 c3vg c4vgveg}; do
  K=$(steps_of $wl)
  B="python bench.py --workload $wl --no-cpu-baseline --no-hbm-resident --no-single-process --multistep 0 --steps 20 --warmup 2 --spinup-ms 0 --repeats 1"
  run 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_${wl}/fetch -- $B > /dev/null 2> $OUT/pmc_${wl}_fetch.err
  run 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_${wl}/write -- $B > /dev/null 2> $OUT/pmc_${wl}_write.err
  run 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH --output-format csv -d $OUT/pmc_${wl}/sq1 -- $B > /dev/null 2> $OUT/pmc_${wl}_sq1.err
  run 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_${wl}/sq2 -- $B > /dev/null 2> $OUT/pmc_${wl}_sq2.err
  python profiles/summarize_pmc.py $OUT/pmc_${wl} $wl $COMMIT > $OUT/pmc_summary_${wl}_fused.json
  cp $OUT/pmc_summary_${wl}_fused.json profiles/$ROUND/
  echo pmc $wl done
  CMD="python bench.py --workload $wl --no-cpu-baseline --no-hbm-resident --no-single-process --multistep 0 --steps $K --warmup 10 --repeats 10"
  run 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$wl -- $CMD > $OUT/bench_trace_$wl.json 2> $OUT/trace_$wl.err
  find $OUT/trace_$wl -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_$wl.csv \;
  run 300 $CMD > $OUT/bench_$wl.json 2> $OUT/bench_$wl.err
  echo trace + bench $wl done
done
# BASELINE config 4 as ONE rank of an 8-GPU strong-scaling run holds it (7 119 columns): the kernel trace shows one launch per step
if [ -z "$WLS" ] || [[ " $WLS " == *" c4 "* ]]; then
  export TRM_BENCH_SHARD_OF=8
  CMD="python bench.py --workload c4 --no-cpu-baseline --no-hbm-resident --no-single-process --multistep 0 --steps 50 --warmup 10 --repeats 10"
  run 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_c4_shard8 -- $CMD > $OUT/bench_trace_c4_shard8.json 2> $OUT/trace_c4_shard8.err
  find $OUT/trace_c4_shard8 -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_c4_shard8.csv \;
  run 300 $CMD > $OUT/bench_c4_shard8.json 2> $OUT/bench_c4_shard8.err
  unset TRM_BENCH_SHARD_OF
  echo trace + bench c4 shard done
fi
[ -n "$WLS" ] && { rm -rf $OUT/pmc_*/ $OUT/trace_*/; exit 0; }      # (WLS="c3 c4": only those workloads again, the rest of $OUT kept)
run 300 python bench.py --integrator heun --no-cpu-baseline --no-hbm-resident --multistep 0 > $OUT/bench_c3_heun.json 2>/dev/null
run 300 python bench.py --kernel unfused --no-cpu-baseline --no-hbm-resident --multistep 0 --repeats 3 > $OUT/bench_c3_unfused.json 2>/dev/null
# the driver's command: headline + HBM-resident companions + multi-step + CPU baseline in one line
run 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
rm -rf $OUT/pmc_*/ $OUT/trace_*/
cut -c1-700 $OUT/bench_default.json
