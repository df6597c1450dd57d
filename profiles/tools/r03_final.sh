# round 3, final GPU call: the whole -m gpu suite, smoke(), the N = 2 rehearsal of bench.py's multi-rank path (two ranks sharing the
# one device over gloo: shape of the line only), deep-column timings
run() { local limit=$1; shift; timeout -k 10 $limit "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: $*"; exit 1; fi; return $rc; }
run 900 python -m pytest tests -m gpu -q -x -W ignore::DeprecationWarning > gpurun_out/final_gpu_tests.log 2>&1; tail -3 gpurun_out/final_gpu_tests.log
run 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/final_smoke.log 2>&1; tail -2 gpurun_out/final_smoke.log
TRM_BENCH_BACKEND=gloo TRM_BENCH_SHARE_DEVICE=1 run 600 python bench.py --gpus 2 --steps 50 --repeats 5 > gpurun_out/rehearsal_gpus2.out 2> gpurun_out/rehearsal_gpus2.err
grep '^{' gpurun_out/rehearsal_gpus2.out | tail -1 > gpurun_out/bench_gpus2_rehearsal.json; cut -c1-300 gpurun_out/bench_gpus2_rehearsal.json
run 300 python profiles/tools/deep_timing.py > gpurun_out/deep_columns_timing.json 2> gpurun_out/deep_timing.err; cut -c1-600 gpurun_out/deep_columns_timing.json
