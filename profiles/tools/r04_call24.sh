# round 4, GPU call 24: the ramp of the HBM-resident step on a fresh box (continuous load for 100 s), then the same after 20 s of idling
timeout -k 10 300 python profiles/tools/ramp_long.py c3x8 100 > gpurun_out/r04_ramp_long.log 2>&1 || exit 1
sleep 20
timeout -k 10 300 python profiles/tools/ramp_long.py c3x8 40 >> gpurun_out/r04_ramp_long.log 2>&1 || exit 1
timeout -k 10 300 python profiles/tools/ramp_long.py c5 60 >> gpurun_out/r04_ramp_long.log 2>&1 || exit 1
grep "^{" gpurun_out/r04_ramp_long.log
