#!/bin/sh
# round 6, call 8: gather_mfma2_kernel (final cut) against gather_mfma_kernel, interleaved repeats on one box; full bf16 gather tests
R=$GRAFT_REPO_ROOT; cd $R
for i in 1 2 3; do
  timeout 300 python tools/gather_c4.py 2>&1 | grep "bf16" | sed -e 's/.*gather_mfma_kernel)://' -e 's/^/v2 /'
  PIPS_LIB_PATH=$R/build/libpips_gmv1.so timeout 300 python tools/gather_c4.py 2>&1 | grep "bf16" | sed -e 's/.*gather_mfma_kernel)://' -e 's/^/v1 /'
done
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_config45_gpu.py -q -k "mfma or bf16 or tiled or dense" 2>&1 | tail -3
