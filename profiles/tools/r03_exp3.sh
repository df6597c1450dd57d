# round 3, GPU call 3: full GPU suite with the new entry points (rows / ring / series window / reset / C host), then the
# two-column-groups kernel A/B
run() { local limit=$1; shift; timeout -k 10 $limit "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: $*"; exit 1; fi; return 0; }
L=gpurun_out/exp3_ab.log; : > $L
run 900 python -m pytest tests -m gpu -q -x -W ignore::DeprecationWarning > gpurun_out/exp3_full.log 2>&1; tail -15 gpurun_out/exp3_full.log
AB="python profiles/tools/ab_options.py"
run 300 $AB c3 g1:column_groups=1 g2:column_groups=2 >> $L 2>&1
run 300 $AB c3x8 g1:column_groups=1 g2:column_groups=2 --steps 60 --reps 5 >> $L 2>&1
run 300 $AB c3vg g1:column_groups=1 g2:column_groups=2 >> $L 2>&1
run 300 $AB c4 g1:column_groups=1 g2:column_groups=2 --steps 50 >> $L 2>&1
run 300 $AB c2n145 g1:column_groups=1 g2:column_groups=2 >> $L 2>&1
cat $L
