# round 4, GPU call 2: the new entry points -- two-call Heun with state-dependent functions, restart through the ABI, one host
# thread driving three contexts, inputs from device memory -- then the whole GPU suite; the coupling-exchange measurement
run() { local limit=$1; shift; timeout -k 10 $limit "$@"; local rc=$?; if [ $rc -ne 0 ]; then echo "FAILED ($rc): $*"; exit 1; fi; return 0; }
python -m pytest tests/test_gpu_state_functions.py tests/test_gpu_restart.py tests/test_gpu_single_process.py -m gpu -q -x -W ignore::DeprecationWarning > gpurun_out/r04_call2_new_tests.log 2>&1
echo "new tests rc=$?"; tail -25 gpurun_out/r04_call2_new_tests.log
run 900 python -m pytest tests -m gpu -q -x -W ignore::DeprecationWarning > gpurun_out/r04_call2_tests.log 2>&1; tail -3 gpurun_out/r04_call2_tests.log
run 600 python profiles/tools/coupling_exchange.py > gpurun_out/r04_coupling_exchange.log 2>&1; cat gpurun_out/r04_coupling_exchange.log | tail -3
