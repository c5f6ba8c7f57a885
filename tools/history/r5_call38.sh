#!/bin/sh
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
PIPS_LIB_PATH=$R/build/libpips_old.so timeout 300 python tools/gather_dump.py /tmp/x_old.pt 2>&1 | grep -v amdgpu | tail -1
PIPS_LIB_PATH=$R/build/libpips_lkp.so timeout 300 python tools/gather_dump.py /tmp/x_lkp.pt 2>&1 | grep -v amdgpu | tail -1
python tools/gather_dump.py --compare /tmp/x_old.pt /tmp/x_lkp.pt | head -3 | tee $O/r5c38_cmp.txt
