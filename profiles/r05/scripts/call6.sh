set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout -k 10 600 python -m pytest tests/test_gpu_surface_in_launch.py -x -q > gpurun_out/r05/call6_tests.log 2>&1
rc=$?
tail -5 gpurun_out/r05/call6_tests.log
[ $rc -ne 0 ] && exit $rc
L=gpurun_out/r05/exp3c_granule_store_forms.log
for v in full frontstore0 frontstore1 frontstore2; do
  if [ $v = full ]; then unset TRM_LIBRARY; else export TRM_LIBRARY=$GRAFT_REPO_ROOT/build/variants/lib_$v.so; fi
  echo "== variant $v" >> $L
  timeout -k 10 200 python profiles/tools/ab_options.py c4 pair:surface_in_launch=0 one:surface_in_launch=1 --steps 50 --reps 5 >> $L 2>&1 || exit 1
done
unset TRM_LIBRARY
timeout -k 10 200 python profiles/tools/ab_options.py c4 pair:surface_in_launch=0 one:surface_in_launch=1 --steps 50 --reps 7 --shard 8 >> $L 2>&1 || exit 1
grep -v amdgpu.ids $L
