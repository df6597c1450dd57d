# A/B of the libraries under build/variants, interleaved, REPS rounds (same box, same clocks)
for r in $(seq ${REPS:-3}); do
for f in build/variants/*.so; do
  TRM_LIBRARY=$PWD/$f python bench.py --no-cpu-baseline --kernel fused "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$f', round(d['roofline']['kernel_ms']*1e3,2), 'us', round(d['roofline']['frac'],3))"
done; done
