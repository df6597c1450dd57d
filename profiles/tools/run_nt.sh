# A/B of the non-temporal load / store variants (build/variants/lib_*.so made by profiles/tools/mkvar.sh)
for wl in c3 c3x8 c5; do
  for v in base ntload ntstore ntboth base; do
    TRM_LIBRARY=$GRAFT_REPO_ROOT/build/variants/lib_$v.so python profiles/tools/ab_simple.py $wl
  done
done
