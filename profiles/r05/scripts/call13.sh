set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
L=gpurun_out/r05/exp4b_in_launch_by_size.log
for sh in 64 16 4; do
  timeout -k 10 300 python profiles/tools/ab_options.py c5 pair:surface_in_launch=0 one:surface_in_launch=1 --steps 50 --reps 5 --shard $sh >> $L 2>&1 || exit 1
done
for sh in 32 2; do
  timeout -k 10 300 python profiles/tools/ab_options.py c4 pair:surface_in_launch=0 one:surface_in_launch=1 --steps 50 --reps 5 --shard $sh >> $L 2>&1 || exit 1
done
timeout -k 10 300 python profiles/tools/ab_options.py c4vg pair:surface_in_launch=0 one:surface_in_launch=1 --steps 50 --reps 5 --shard 8 >> $L 2>&1 || exit 1
grep -v amdgpu.ids $L
