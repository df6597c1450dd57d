# round 3, GPU call 6: interleaved LandModel launches (k_land_euler / k_land_pk) -- tests, A/B
run() { local limit=$1; shift; timeout -k 10 $limit "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: $*"; exit 1; fi; return 0; }
L=gpurun_out/exp6_ab.log; : > $L
run 900 python -m pytest tests -m gpu -q -x -W ignore::DeprecationWarning > gpurun_out/exp6_full.log 2>&1; tail -5 gpurun_out/exp6_full.log
AB="python profiles/tools/ab_options.py"
run 300 $AB c4 off:pipeline_parts=0 on:pipeline_parts=1 --steps 50 >> $L 2>&1
run 300 $AB c4vg off:pipeline_parts=0 on:pipeline_parts=1 --steps 50 >> $L 2>&1
run 300 $AB c5 off:pipeline_parts=0 on:pipeline_parts=1 --steps 40 --reps 5 >> $L 2>&1
run 300 $AB c5vg off:pipeline_parts=0 on:pipeline_parts=1 --steps 40 --reps 5 >> $L 2>&1
run 300 $AB c4 off:pipeline_parts=0 on:pipeline_parts=1 --shard 8 --steps 50 >> $L 2>&1
run 300 $AB c4 off:pipeline_parts=0 on:pipeline_parts=1 --shard 2 --steps 50 >> $L 2>&1
cat $L
