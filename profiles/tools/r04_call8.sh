# round 4, GPU call 8: the C3 rows of the collection again without the single-process leg in the traced command (its time-sliced
# launches of the same kernel had raised the CSV average), and the fp32 memory floors by access shape
COMMIT=$1
mkdir -p gpurun_out/prof_r04
WLS="c3" bash profiles/collect.sh r04 $COMMIT || exit 1
hipcc --offload-arch=gfx950 -O3 -o /tmp/memfloor_f32 profiles/tools/microbench/memfloor_f32.hip || exit 1
timeout -k 10 120 /tmp/memfloor_f32 > gpurun_out/r04_memfloor_f32.log 2>&1 || exit 1
timeout -k 10 120 /tmp/memfloor_f32 >> gpurun_out/r04_memfloor_f32.log 2>&1 || exit 1
cat gpurun_out/r04_memfloor_f32.log
