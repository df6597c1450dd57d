# round 4, GPU call 6: columns of 129 ... 256 levels (four levels per lane) -- tests; deep / wide timings against the reference-order
# kernels (profiles/r04/deep_columns_timing.json); then the default bench line (shape check of the single-process leg)
run() { local limit=$1; shift; timeout -k 10 $limit "$@"; local rc=$?; if [ $rc -ne 0 ]; then echo "FAILED ($rc): $*"; exit 1; fi; return 0; }
run 900 python -m pytest tests/test_gpu_deep_columns.py tests/test_gpu_parity.py -m gpu -q -x -W ignore::DeprecationWarning > gpurun_out/r04_call6_tests.log 2>&1; tail -3 gpurun_out/r04_call6_tests.log
run 900 python profiles/tools/deep_timing.py N145 > gpurun_out/r04_deep_columns_timing.json 2> gpurun_out/r04_deep_timing.err; cut -c1-3000 gpurun_out/r04_deep_columns_timing.json
run 600 python bench.py > gpurun_out/r04_bench_default_probe.json 2> gpurun_out/r04_bench_default_probe.err; cut -c1-1500 gpurun_out/r04_bench_default_probe.json
