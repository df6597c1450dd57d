# round 3, first GPU call: the new defaults (resident program for nsteps > 1, two-part pipeline) -- tests, then A/B
# A step that runs into its time limit ends the script (no further GPU work after a hang).
run() { local limit=$1; shift; timeout -k 10 $limit "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: $*"; exit 1; fi; return 0; }
L=gpurun_out/exp1_ab.log; : > $L
run 300 python -m pytest tests/test_gpu_column_programs.py -q -x -W ignore::DeprecationWarning > gpurun_out/exp1_tests.log 2>&1; tail -3 gpurun_out/exp1_tests.log
AB="python profiles/tools/ab_options.py"
run 200 $AB c4 off:pipeline_parts=0 on:pipeline_parts=1 >> $L 2>&1
run 200 $AB c4vg off:pipeline_parts=0 on:pipeline_parts=1 >> $L 2>&1
run 200 $AB c4vgveg off:pipeline_parts=0 on:pipeline_parts=1 >> $L 2>&1
run 200 $AB c3 off:pipeline_parts=0 on:pipeline_parts=1 multi:steps_per_launch=0 >> $L 2>&1
run 200 $AB c3x8 off:pipeline_parts=0 on:pipeline_parts=1 --steps 60 --reps 5 >> $L 2>&1
run 300 $AB c5 off:pipeline_parts=0 on:pipeline_parts=1 multi:steps_per_launch=50,pipeline_parts=0 --steps 50 --reps 5 >> $L 2>&1
run 300 $AB c5vg off:pipeline_parts=0 on:pipeline_parts=1 --steps 50 --reps 5 >> $L 2>&1
run 200 $AB c4 off:pipeline_parts=0 on:pipeline_parts=1 multi:steps_per_launch=0 --shard 8 --steps 200 >> $L 2>&1
run 200 $AB c2 per:steps_per_launch=1 multi:steps_per_launch=0 --steps 200 >> $L 2>&1
cat $L
run 600 python -m pytest tests -m gpu -q -x -W ignore::DeprecationWarning > gpurun_out/exp1_full.log 2>&1; tail -3 gpurun_out/exp1_full.log
