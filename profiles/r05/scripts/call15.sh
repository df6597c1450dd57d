set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
L=gpurun_out/r05/exp6_c2_bisect.log
for rep in 1 2 3; do
  for lib in r04 c01e530 06062aa 3de9757 head; do
    if [ $lib = head ]; then unset TRM_LIBRARY; else export TRM_LIBRARY=$GRAFT_REPO_ROOT/build/variants/lib_$lib.so; fi
    echo -n "$lib " >> $L
    timeout -k 10 300 python profiles/tools/ab_options.py c2 x: --steps 100 --reps 7 2>/dev/null | grep workload | cut -c1-200 >> $L || exit 1
  done
done
cat $L
