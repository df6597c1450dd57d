# round 3, GPU call 5: full GPU suite with the deep-column kernel, its timing, the driver's bench line, the N > 1 rehearsal
run() { local limit=$1; shift; timeout -k 10 $limit "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: $*"; exit 1; fi; return 0; }
run 900 python -m pytest tests -m gpu -q -x -W ignore::DeprecationWarning > gpurun_out/exp5_full.log 2>&1; tail -5 gpurun_out/exp5_full.log
run 300 python profiles/tools/deep_timing.py > gpurun_out/exp5_deep.json 2>&1; cat gpurun_out/exp5_deep.json
run 600 python bench.py > gpurun_out/exp5_bench_default.json 2> gpurun_out/exp5_bench_default.err; cut -c1-300 gpurun_out/exp5_bench_default.json
TRM_BENCH_BACKEND=gloo TRM_BENCH_SHARE_DEVICE=1 run 600 python bench.py --gpus 2 --steps 50 --repeats 5 > gpurun_out/exp5_bench_gpus2_rehearsal.json 2> gpurun_out/exp5_bench_gpus2_rehearsal.err; cut -c1-3000 gpurun_out/exp5_bench_gpus2_rehearsal.json; tail -3 gpurun_out/exp5_bench_gpus2_rehearsal.err
