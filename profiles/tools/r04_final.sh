# round 4, final check on the committed build: the whole GPU suite, smoke(), the driver's bench command
run() { local limit=$1; shift; timeout -k 10 $limit "$@"; local rc=$?; if [ $rc -ne 0 ]; then echo "FAILED ($rc): $*"; exit 1; fi; return 0; }
run 1100 python -m pytest tests -m gpu -q -x -W ignore::DeprecationWarning > gpurun_out/r04_final_tests.log 2>&1; tail -3 gpurun_out/r04_final_tests.log
run 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r04_final_smoke.log 2>&1; tail -2 gpurun_out/r04_final_smoke.log
run 600 python bench.py > gpurun_out/r04_final_bench.json 2> gpurun_out/r04_final_bench.err; cut -c1-900 gpurun_out/r04_final_bench.json
