#!/bin/sh
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 300 python tools/gather_dump.py /tmp/x_prod.pt 2>&1 | grep -v amdgpu | tail -1
PIPS_LIB_PATH=$R/build/libpips_wave.so timeout 300 python tools/gather_dump.py /tmp/x_wave.pt 2>&1 | grep -v amdgpu | tail -1
python tools/gather_dump.py --compare /tmp/x_prod.pt /tmp/x_wave.pt | tee $O/r5c32_wave_vs_block.txt
PIPS_LIB_PATH=$R/build/libpips_wave.so timeout 300 python -u tools/gather_c4.py 2>&1 | grep -v amdgpu | grep "bf16 mode" | tee -a $O/r5c32_wave_vs_block.txt
