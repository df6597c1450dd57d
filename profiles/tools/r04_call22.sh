# round 4, GPU call 22: the memory floor of the HBM-resident step (3 reads + 6 writes per cell) on the shipped layout (one buffer per
# field) against a tiled layout (the six fields of 8 columns adjacent), 8 x N145 and N145
hipcc --offload-arch=gfx950 -O3 -o /tmp/memfloor profiles/tools/microbench/memfloor.hip || exit 1
for n in 455608 56951; do timeout -k 10 200 /tmp/memfloor $n || exit 1; done > gpurun_out/r04_memfloor_tiled.log 2>&1
cat gpurun_out/r04_memfloor_tiled.log
