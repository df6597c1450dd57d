#!/bin/bash
# rehearsal of the N > 1 bench path on the one GPU: two ranks share the device, gloo for the host-side reductions
set -o pipefail
mkdir -p gpurun_out
TRM_BENCH_BACKEND=gloo TRM_BENCH_SHARE_DEVICE=1 timeout -k 10 600 python bench.py --gpus 2 --steps 20 --warmup 5 --cpu-seconds 3 > gpurun_out/bench_world2_rehearsal.json 2> gpurun_out/bench_world2_rehearsal.err
rc=$?
tail -3 gpurun_out/bench_world2_rehearsal.err
cut -c1-1500 gpurun_out/bench_world2_rehearsal.json
exit $rc
