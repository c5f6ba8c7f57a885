#!/bin/sh
# round 6, call 9: gather_mfma_kernel with the round's three local changes (12-state guard, batched fragment reads, column blend with
# contiguous stores) against its round-5 build and gather_mfma2_kernel: bits and time
R=$GRAFT_REPO_ROOT; cd $R
PIPS_LIB_PATH=$R/build/libpips_gmv1new.so timeout 300 python tools/gather_dump.py /tmp/v1n.pt 2>&1 | tail -1
PIPS_LIB_PATH=$R/build/libpips_gmv1.so timeout 300 python tools/gather_dump.py /tmp/v1.pt 2>&1 | tail -1
python tools/gather_dump.py --compare /tmp/v1.pt /tmp/v1n.pt 2>&1 | head -3
for i in 1 2; do
  PIPS_LIB_PATH=$R/build/libpips_gmv1new.so timeout 300 python tools/gather_c4.py 2>&1 | grep "bf16\|config-3" | sed -e 's/.*gather_mfma_kernel)://' -e 's/^/v1new /'
  PIPS_LIB_PATH=$R/build/libpips_gmv1.so timeout 300 python tools/gather_c4.py 2>&1 | grep "bf16" | sed -e 's/.*gather_mfma_kernel)://' -e 's/^/v1 /'
done
