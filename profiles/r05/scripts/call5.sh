set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
L=gpurun_out/r05/exp3b_surface_in_launch_diag.log
for v in full frontdiag6 frontdiag7 frontw1; do
  if [ $v = full ]; then unset TRM_LIBRARY; else export TRM_LIBRARY=$GRAFT_REPO_ROOT/build/variants/lib_$v.so; fi
  echo "== variant $v" >> $L
  timeout -k 10 200 python profiles/tools/ab_options.py c4 pair:surface_in_launch=0 one:surface_in_launch=1 --steps 50 --reps 5 >> $L 2>&1 || exit 1
done
grep -v amdgpu.ids $L
