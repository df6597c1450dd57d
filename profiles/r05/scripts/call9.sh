set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout -k 10 900 python -m pytest tests/test_gpu_program_selection.py tests/test_gpu_output_ring.py -x -q > gpurun_out/r05/call9_tests.log 2>&1
rc=$?
tail -12 gpurun_out/r05/call9_tests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python bench.py --no-cpu-baseline --no-hbm-resident --no-single-process --multistep 0 > gpurun_out/r05/call9_bench_c3.json 2> gpurun_out/r05/call9_bench_c3.err || { tail -5 gpurun_out/r05/call9_bench_c3.err; exit 1; }
timeout -k 10 300 python bench.py --workload c4 --steps 50 --no-cpu-baseline --no-hbm-resident --no-single-process --multistep 0 > gpurun_out/r05/call9_bench_c4.json 2> gpurun_out/r05/call9_bench_c4.err || { tail -5 gpurun_out/r05/call9_bench_c4.err; exit 1; }
python - <<'PY'
import json
for wl in ("c3", "c4"):
    d = json.load(open(f"gpurun_out/r05/call9_bench_{wl}.json"))
    print(wl, "ms_per_step", d["ms_per_step"], "kernel_ms", d["roofline"]["kernel_ms"], "frac", round(d["roofline"]["frac"], 4), d["roofline"]["kernel"], d["sustained"])
PY
