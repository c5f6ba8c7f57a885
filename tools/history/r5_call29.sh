#!/bin/sh
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
PIPS_LIB_PATH=$R/build/libpips_wave.so timeout 600 python -m pytest tests -m gpu -x -q -k "mfma or bf16_matrix" 2>&1 | tail -15 | tee $O/r5c29_tests.txt
PIPS_LIB_PATH=$R/build/libpips_wave.so timeout 300 python -u tools/gather_c4.py 2>&1 | grep -v amdgpu | grep "bf16 mode" | tee $O/r5c29_gather.txt
