# round 3, GPU call 7: full GPU suite on the final kernels, deep-column timing with the derivation, the driver's bench line
run() { local limit=$1; shift; timeout -k 10 $limit "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: $*"; exit 1; fi; return 0; }
run 900 python -m pytest tests -m gpu -q -x -W ignore::DeprecationWarning > gpurun_out/exp7_full.log 2>&1; tail -5 gpurun_out/exp7_full.log
run 300 python profiles/tools/deep_timing.py > gpurun_out/exp7_deep.json 2>&1; cat gpurun_out/exp7_deep.json
run 600 python bench.py > gpurun_out/exp7_bench_default.json 2> gpurun_out/exp7_bench_default.err; python -c "
import json; d=json.load(open('gpurun_out/exp7_bench_default.json')); h=d['roofline_hbm_resident']; c=d['cpu_baseline']
print('c3', d['roofline']['frac'], d['ms_per_step'], 'c3x8', h['frac'], 'c5', h['also'][0]['frac'], h['also'][0].get('land_interleaved'))
print('multi', d['multistep']['us_per_step']); print('cpu', c['value'], c['cores'], c.get('streamed_GBps'), c.get('thread_scan'), c.get('cgroup_cpu_quota'), c.get('hardware_threads'))
print(c['sample'][:600])"
