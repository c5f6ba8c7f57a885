#!/bin/sh
# round 6, call 3: gather_mfma2_kernel (LDS-DMA, blend in the aux waves) against the round-5 kernel: bits and time
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 300 python tools/gather_dump.py /tmp/v2.pt 2>&1 | tail -1
PIPS_LIB_PATH=$R/build/libpips_gmv1.so timeout 300 python tools/gather_dump.py /tmp/v1.pt 2>&1 | tail -1
python tools/gather_dump.py --compare /tmp/v1.pt /tmp/v2.pt 2>&1 | head -12
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "mfma or bf16_maps or tiled" 2>&1 | tail -5
timeout 300 python tools/gather_c4.py 2>&1 | grep "bf16\|config-3"
PIPS_LIB_PATH=$R/build/libpips_gmv1.so timeout 300 python tools/gather_c4.py 2>&1 | grep "bf16"
