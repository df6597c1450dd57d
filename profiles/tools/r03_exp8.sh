# round 3, GPU call 8: workgroup size of the step kernels (TRM_STEP_BLOCK = 64 / 128 / 256 (shipped) / 512), one process per variant
run() { local limit=$1; shift; timeout -k 10 $limit "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: $*"; exit 1; fi; return 0; }
L=gpurun_out/exp8_ab.log; : > $L
AB="python profiles/tools/ab_options.py"
for wl in c3x8 c5 c3 c4; do
  case $wl in c3x8) S="--steps 60 --reps 5";; c5) S="--steps 30 --reps 5";; c4) S="--steps 50";; *) S="";; esac
  for B in 256 64 128 512 256; do
    if [ $B = 256 ]; then unset TRM_LIBRARY; else export TRM_LIBRARY=$PWD/build/variants/libtrm_block$B.so; fi
    echo "block $B" >> $L
    run 300 $AB $wl b$B: $S >> $L 2>&1
  done
done
cat $L
