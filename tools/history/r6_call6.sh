#!/bin/sh
# round 6, call 6: which part of the product step collides with the LDS-DMA stream (probes on top of "no blend": 8)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
for a in 8 24 40 56 9 25 41; do
  sh tools/build_gather_variant.sh g2a$a -DG2_ABLATE=$a > /dev/null 2>&1
  echo "== G2_ABLATE=$a"
  PIPS_LIB_PATH=$R/build/libpips_g2a$a.so timeout 300 python tools/gather_c4.py 2>&1 | grep "bf16" | sed -e 's/.*gather_mfma_kernel)://' | head -1
done
