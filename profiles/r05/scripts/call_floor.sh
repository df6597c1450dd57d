#!/bin/bash
# the launch floor: period of back-to-back dependent launches by grid size and argument size; with kernel arguments in host memory too
set -o pipefail
mkdir -p gpurun_out
hipcc --offload-arch=gfx950 -O3 -o /tmp/launchfloor profiles/tools/microbench/launchfloor.hip 2>/dev/null || exit 1
L=gpurun_out/exp13_launch_floor.log
echo "== default" > $L
timeout -k 10 200 /tmp/launchfloor >> $L 2>&1 || exit 1
echo "== HIP_FORCE_DEV_KERNARG=0" >> $L
HIP_FORCE_DEV_KERNARG=0 timeout -k 10 200 /tmp/launchfloor >> $L 2>&1 || exit 1
echo "== HIP_FORCE_DEV_KERNARG=1" >> $L
HIP_FORCE_DEV_KERNARG=1 timeout -k 10 200 /tmp/launchfloor >> $L 2>&1 || exit 1
cat $L
