python -m pytest tests/test_gpu_parity.py tests/test_gpu_column_programs.py tests/test_gpu_reference_tests.py -q -x > gpurun_out/pk.log 2>&1; tail -3 gpurun_out/pk.log
for wl in c3vg c4vg c5vg; do python bench.py --workload $wl --no-cpu-baseline --no-hbm-resident --multistep 0 --steps 50 --warmup 5 > gpurun_out/$wl.json 2> gpurun_out/$wl.err; python -c "
import json; d=json.load(open('gpurun_out/$wl.json')); print('$wl', round(d['value']/1e9,3), d['roofline']['kernel_ms'], round(d['roofline']['frac'],3))"; done
