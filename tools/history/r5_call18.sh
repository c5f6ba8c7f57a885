#!/bin/sh
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 600 python -m pytest tests -m gpu -x -q -k "mfma or bf16_matrix" 2>&1 | tail -2
timeout 300 python -u tools/gather_c4.py 2>&1 | grep -v amdgpu | tee $O/r5c18_gather.txt
