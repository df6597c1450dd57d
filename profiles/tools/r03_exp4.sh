# round 3, GPU call 4: full GPU suite after the kernel clean-up, liquid-fraction-only derivation A/B, the driver's bench line
run() { local limit=$1; shift; timeout -k 10 $limit "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: $*"; exit 1; fi; return 0; }
L=gpurun_out/exp4_ab.log; : > $L
run 900 python -m pytest tests -m gpu -q -x -W ignore::DeprecationWarning > gpurun_out/exp4_full.log 2>&1; tail -5 gpurun_out/exp4_full.log
AB="python profiles/tools/ab_options.py"
D="none:derive_closure_fields=0 both:derive_closure_fields=1 liq:derive_closure_fields=3"
run 300 $AB c3 $D >> $L 2>&1
run 300 $AB c3x8 $D --steps 60 --reps 5 >> $L 2>&1
run 300 $AB c4 $D --steps 50 >> $L 2>&1
run 300 $AB c4vg $D --steps 50 >> $L 2>&1
run 300 $AB c5 none:derive_closure_fields=0 liq:derive_closure_fields=3 --steps 40 --reps 5 >> $L 2>&1
run 300 $AB c5vg none:derive_closure_fields=0 liq:derive_closure_fields=3 --steps 40 --reps 5 >> $L 2>&1
cat $L
run 600 python bench.py > gpurun_out/exp4_bench_default.json 2> gpurun_out/exp4_bench_default.err; cut -c1-1500 gpurun_out/exp4_bench_default.json; tail -3 gpurun_out/exp4_bench_default.err
