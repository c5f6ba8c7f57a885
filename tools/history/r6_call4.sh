#!/bin/sh
# round 6, call 4: timing probes of gather_mfma2_kernel (G2_ABLATE: 1 no DMA, 2 no products, 4 no tap stores, 8 no blend)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
for a in 1 2 4 8 9 10 12 14; do
  sh tools/build_gather_variant.sh g2a$a -DG2_ABLATE=$a > /dev/null 2>&1
  echo "== G2_ABLATE=$a"
  PIPS_LIB_PATH=$R/build/libpips_g2a$a.so timeout 300 python tools/gather_c4.py 2>&1 | grep "bf16" | sed -e 's/.*gather_mfma_kernel)://'
done
