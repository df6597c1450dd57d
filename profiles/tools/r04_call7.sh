# round 4, GPU call 7: the round's profile collection (profiles/collect.sh r04: PMC passes, one rocprof kernel-trace CSV per workload,
# the bench lines) and the N = 2 rehearsal of bench.py's multi-rank line on the one GPU (gloo, shared device)
COMMIT=$1
bash profiles/collect.sh r04 $COMMIT || exit 1
OUT=gpurun_out/prof_r04
TRM_BENCH_BACKEND=gloo TRM_BENCH_SHARE_DEVICE=1 timeout -k 10 600 python bench.py --gpus 2 --steps 50 --repeats 3 > $OUT/bench_gpus2_rehearsal.json 2> $OUT/bench_gpus2_rehearsal.err || { echo "FAILED: rehearsal"; tail -5 $OUT/bench_gpus2_rehearsal.err; exit 1; }
cut -c1-600 $OUT/bench_gpus2_rehearsal.json
