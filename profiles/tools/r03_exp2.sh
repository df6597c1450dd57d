# round 3, GPU call 2: (a) first step at which the synthetic states raise a status flag, (b) L2 prefetch-ahead A/B,
# (c) two-part pipeline with a single context in the process and more hardware queues
run() { local limit=$1; shift; timeout -k 10 $limit "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: $*"; exit 1; fi; return 0; }
L=gpurun_out/exp2_ab.log; : > $L
run 300 python -m pytest tests/test_gpu_column_programs.py tests/test_gpu_parity.py -q -x -W ignore::DeprecationWarning > gpurun_out/exp2_tests.log 2>&1; tail -3 gpurun_out/exp2_tests.log
run 300 python profiles/tools/first_flag.py c3 c3vg c4 c4vg c5 >> $L 2>&1
AB="python profiles/tools/ab_options.py"
P="p0:prefetch_columns=0 p8k:prefetch_columns=8192 p16k:prefetch_columns=16384 p32k:prefetch_columns=32768 p64k:prefetch_columns=65536"
run 300 $AB c3x8 $P --steps 60 --reps 5 >> $L 2>&1
run 300 $AB c5 $P --steps 40 --reps 5 >> $L 2>&1
run 300 $AB c3 p0:prefetch_columns=0 p4k:prefetch_columns=4096 p8k:prefetch_columns=8192 p16k:prefetch_columns=16384 >> $L 2>&1
run 300 $AB c4 p0:prefetch_columns=0 p8k:prefetch_columns=8192 p16k:prefetch_columns=16384 --steps 50 >> $L 2>&1
run 300 $AB c4vg p0:prefetch_columns=0 p16k:prefetch_columns=16384 --steps 50 >> $L 2>&1
run 300 $AB c5vg p0:prefetch_columns=0 p16k:prefetch_columns=16384 p32k:prefetch_columns=32768 --steps 40 --reps 5 >> $L 2>&1
run 200 $AB c4 on:pipeline_parts=1 --steps 50 >> $L 2>&1
GPU_MAX_HW_QUEUES=8 run 200 $AB c4 on:pipeline_parts=1 off:pipeline_parts=0 --steps 50 >> $L 2>&1
cat $L
