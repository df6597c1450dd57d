set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
L=gpurun_out/r05/exp5_round4_library_vs_head.log
for rep in 1 2 3; do
  for wl in c3 c3x8 c4 c5 c2; do
    steps=50; [ $wl = c3 ] && steps=100; [ $wl = c2 ] && steps=100
    for lib in r04 head; do
      if [ $lib = head ]; then unset TRM_LIBRARY; else export TRM_LIBRARY=$GRAFT_REPO_ROOT/build/variants/lib_r04.so; fi
      echo -n "$lib " >> $L
      timeout -k 10 300 python profiles/tools/ab_options.py $wl x: --steps $steps --reps 5 2>/dev/null | grep workload >> $L || exit 1
    done
  done
done
cat $L
