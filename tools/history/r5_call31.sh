#!/bin/sh
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 600 python -m pytest tests -m gpu -x -q -k "mfma or bf16_matrix" 2>&1 | tail -3 | tee $O/r5c31_tests.txt
PIPS_LIB_PATH=$R/build/libpips_wave.so timeout 600 python -m pytest tests -m gpu -x -q -k "mfma or bf16_matrix" 2>&1 | tail -3 | tee -a $O/r5c31_tests.txt
timeout 300 python -u tools/gather_c4.py 2>&1 | grep -v amdgpu | grep "bf16 mode" | tee $O/r5c31_gather.txt
echo "== wave kernel" | tee -a $O/r5c31_gather.txt
PIPS_LIB_PATH=$R/build/libpips_wave.so timeout 300 python -u tools/gather_c4.py 2>&1 | grep -v amdgpu | grep "bf16 mode" | tee -a $O/r5c31_gather.txt
