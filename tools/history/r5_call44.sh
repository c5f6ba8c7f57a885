#!/bin/sh
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 300 python tools/gather_dump.py /tmp/x_prod.pt 2>&1 | grep -v amdgpu | tail -1
PIPS_LIB_PATH=$R/build/libpips_lk7.so timeout 300 python tools/gather_dump.py /tmp/x_lk7.pt 2>&1 | grep -v amdgpu | tail -1
python tools/gather_dump.py --compare /tmp/x_prod.pt /tmp/x_lk7.pt | head -2 | tee $O/r5c44_lookup_fixed.txt
timeout 900 python -m pytest tests -m gpu -x -q -k "mfma or bf16_matrix or batches or config4" 2>&1 | tail -3 | tee $O/r5c44_tests.txt
timeout 300 python -u tools/gather_c4.py 2>&1 | grep -v amdgpu | grep "bf16 mode" | tee $O/r5c44_gather.txt
PIPS_LIB_PATH=$R/build/libpips_lk7.so timeout 300 python -u tools/gather_c4.py 2>&1 | grep -v amdgpu | grep "bf16 mode" | tee -a $O/r5c44_gather.txt
