set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout -k 10 900 python -m pytest tests/test_gpu_surface_in_launch.py tests/test_gpu_program_selection.py tests/test_gpu_column_programs.py -x -q > gpurun_out/r05/call18_tests.log 2>&1
rc=$?
tail -12 gpurun_out/r05/call18_tests.log
[ $rc -ne 0 ] && exit $rc
L=gpurun_out/r05/exp8_heun_surface_in_launch.log
timeout -k 10 300 python profiles/tools/ab_options.py c4 pair:surface_in_launch=0 one:surface_in_launch=1 --steps 50 --reps 5 --heun >> $L 2>&1 || exit 1
timeout -k 10 300 python profiles/tools/ab_options.py c4 pair:surface_in_launch=0 one:surface_in_launch=1 --steps 50 --reps 5 --heun --shard 8 >> $L 2>&1 || exit 1
timeout -k 10 300 python profiles/tools/ab_options.py c4vg pair:surface_in_launch=0 one:surface_in_launch=1 --steps 50 --reps 5 --heun >> $L 2>&1 || exit 1
grep -v amdgpu.ids $L
