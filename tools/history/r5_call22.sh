#!/bin/sh
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 600 python -m pytest tests -m gpu -x -q -k "mfma or bf16_matrix" 2>&1 | tail -2
timeout 300 python -u tools/gather_c4.py 2>&1 | grep -v amdgpu | grep "bf16 mode\|fp32 (" | tee $O/r5c22_gather.txt
PIPS_LIB_PATH=$R/build/libpips_trace.so timeout 200 python tools/gm_trace.py 2>&1 | grep -v amdgpu.ids | head -28 > $O/r5c22_trace.txt
cat $O/r5c22_trace.txt
