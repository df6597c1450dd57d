# round 4, GPU call 20: the whole GPU suite on the current build; the one-launch Heun step with the boundary signature compiled in
# against the run-time kinds (both in one process); then the round's profile collection on this build
COMMIT=$1
run() { local limit=$1; shift; timeout -k 10 $limit "$@"; local rc=$?; if [ $rc -ne 0 ]; then echo "FAILED ($rc): $*"; tail -40 gpurun_out/r04_call20_tests.log; exit 1; fi; return 0; }
run 1100 python -m pytest tests -m gpu -q -x -W ignore::DeprecationWarning > gpurun_out/r04_call20_tests.log 2>&1; tail -3 gpurun_out/r04_call20_tests.log
L=gpurun_out/r04_exp15_heun_signature.log; : > $L
AB="python profiles/tools/ab_options.py"
for round in 1 2 3; do
  run 300 $AB c3 runtime:bc_signature=0 compiled:bc_signature=1 --heun --steps 100 --reps 7 >> $L 2>&1
  run 300 $AB c4 runtime:bc_signature=0 compiled:bc_signature=1 --heun --steps 50 --reps 7 >> $L 2>&1
  run 300 $AB c2 runtime:bc_signature=0 compiled:bc_signature=1 --heun --steps 200 --reps 7 >> $L 2>&1
done
grep -h "^{" $L | cut -c1-300
bash profiles/collect.sh r04 $COMMIT || exit 1
