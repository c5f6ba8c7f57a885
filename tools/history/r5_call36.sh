#!/bin/sh
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 900 python -m pytest tests -m gpu -x -q -k "mfma or bf16_matrix or batches" 2>&1 | grep -v amdgpu | tail -40 | tee $O/r5c36_tests.txt
timeout 300 python tools/gather_dump.py /tmp/x_prod.pt 2>&1 | grep -v amdgpu | tail -1
