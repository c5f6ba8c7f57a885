#!/bin/sh
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
PIPS_LIB_PATH=$R/build/libpips_lkp.so timeout 300 python -u tools/gather_c4.py 2>&1 | grep -v amdgpu | grep "bf16 mode" | tee $O/r5c39_gather.txt
