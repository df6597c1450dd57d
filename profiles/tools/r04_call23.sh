# round 4, GPU call 23: the step kernels against their OWN memory-only floors on ONE box in one call: memfloor (3 reads + 6 writes per
# cell, fp64, 8 x N145 and N145), memfloor_f32 (C5's ten accesses), then the steps themselves (one process each, three rounds)
hipcc --offload-arch=gfx950 -O3 -o /tmp/memfloor profiles/tools/microbench/memfloor.hip || exit 1
hipcc --offload-arch=gfx950 -O3 -o /tmp/memfloor_f32 profiles/tools/microbench/memfloor_f32.hip || exit 1
L=gpurun_out/r04_step_vs_floor.log; : > $L
for round in 1 2 3; do
  for n in 455608 56951; do timeout -k 10 200 /tmp/memfloor $n 2>&1 | grep -E "^D|^E" >> $L || exit 1; done
  timeout -k 10 200 /tmp/memfloor_f32 2>&1 | grep -E "^P " >> $L || exit 1
  for wl in c3x8 c3 c5; do timeout -k 10 300 python profiles/tools/ab_options.py $wl shipped: --steps 60 --reps 5 2>&1 | grep "^{" | cut -c1-200 >> $L || exit 1; done
done
cat $L
