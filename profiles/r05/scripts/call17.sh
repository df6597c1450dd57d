set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
echo "== TRM_STAGED_SMALL=1 TRM_SCALAR_INPUTS=0 TRM_DERIVE_DEFAULT=1" > gpurun_out/r05/call17_forced_variants.log
TRM_STAGED_SMALL=1 TRM_SCALAR_INPUTS=0 TRM_DERIVE_DEFAULT=1 timeout -k 10 900 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_program_selection.py 2>&1 | tail -4 >> gpurun_out/r05/call17_forced_variants.log || { tail -30 gpurun_out/r05/call17_forced_variants.log; exit 1; }
echo "== TRM_DERIVE_DEFAULT=1" >> gpurun_out/r05/call17_forced_variants.log
TRM_DERIVE_DEFAULT=1 timeout -k 10 900 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_program_selection.py 2>&1 | tail -4 >> gpurun_out/r05/call17_forced_variants.log || { tail -30 gpurun_out/r05/call17_forced_variants.log; exit 1; }
echo "== TRM_STAGED_SMALL=1 TRM_SCALAR_INPUTS=1 TRM_DERIVE_DEFAULT=1" >> gpurun_out/r05/call17_forced_variants.log
TRM_STAGED_SMALL=1 TRM_SCALAR_INPUTS=1 TRM_DERIVE_DEFAULT=1 timeout -k 10 900 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_program_selection.py 2>&1 | tail -4 >> gpurun_out/r05/call17_forced_variants.log || { tail -30 gpurun_out/r05/call17_forced_variants.log; exit 1; }
cat gpurun_out/r05/call17_forced_variants.log
