set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
L=gpurun_out/r05/exp2b_tail_diag.log
for v in 1 2 3 3s1 63; do
  export TRM_LIBRARY=$GRAFT_REPO_ROOT/build/variants/lib_taildiag$v.so
  echo "== variant $v" >> $L
  timeout -k 10 200 python profiles/tools/ab_options.py c4 pair:tail_surface=0 tail:tail_surface=1 --steps 50 --reps 5 >> $L 2>&1 || exit 1
done
grep -v amdgpu.ids $L
