#!/bin/sh
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 300 python tools/gather_dump.py /tmp/x_prod.pt 2>&1 | grep -v amdgpu | tail -1
PIPS_LIB_PATH=$R/build/libpips_old.so timeout 300 python tools/gather_dump.py /tmp/x_old.pt 2>&1 | grep -v amdgpu | tail -1
python tools/gather_dump.py --compare /tmp/x_old.pt /tmp/x_prod.pt | head -3 | tee $O/r5c46_cmp.txt
timeout 900 python -m pytest tests -m gpu -x -q -k "mfma or bf16_matrix or batches or repeatable or config4 or tiled" 2>&1 | tail -3 | tee $O/r5c46_tests.txt
timeout 300 python -u tools/gather_c4.py 2>&1 | grep -v amdgpu | grep "bf16 mode\|fp32 (\|config-3" | tee $O/r5c46_gather.txt
PIPS_LIB_PATH=$R/build/libpips_trace.so timeout 200 python tools/gm_trace.py 2>&1 | grep -v amdgpu.ids | head -26 > $O/r5c46_trace.txt
cat $O/r5c46_trace.txt
