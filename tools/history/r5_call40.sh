#!/bin/sh
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 900 python -m pytest tests -m gpu -x -q -k "mfma or bf16_matrix or batches or config4" 2>&1 | tail -3 | tee $O/r5c40_tests.txt
timeout 300 python -u tools/gather_c4.py 2>&1 | grep -v amdgpu | grep "bf16 mode\|config-3" | tee $O/r5c40_gather.txt
