# round 4, GPU call 33: the whole GPU suite and the round's profile collection on the final commit
COMMIT=$1
run() { local limit=$1; shift; timeout -k 10 $limit "$@"; local rc=$?; if [ $rc -ne 0 ]; then echo "FAILED ($rc): $*"; tail -40 gpurun_out/r04_call33_tests.log; exit 1; fi; return 0; }
run 1100 python -m pytest tests -m gpu -q -x -W ignore::DeprecationWarning > gpurun_out/r04_call33_tests.log 2>&1; tail -3 gpurun_out/r04_call33_tests.log
bash profiles/collect.sh r04 $COMMIT || exit 1
