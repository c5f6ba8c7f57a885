#!/bin/sh
# round 6, call 7: per-phase clocks of gather_mfma2_kernel (register-accumulated trace), full kernel and without the blend
R=$GRAFT_REPO_ROOT; cd $R
sh tools/build_gather_variant.sh g2tr -DG2_TRACE > /dev/null 2>&1
PIPS_LIB_PATH=$R/build/libpips_g2tr.so timeout 300 python tools/g2_trace.py 2>&1 | grep -v amdgpu.ids
sh tools/build_gather_variant.sh g2tr8 -DG2_TRACE -DG2_ABLATE=8 > /dev/null 2>&1
echo "== no blend"; PIPS_LIB_PATH=$R/build/libpips_g2tr8.so timeout 300 python tools/g2_trace.py 2>&1 | grep -v amdgpu.ids
sh tools/build_gather_variant.sh g2tr10 -DG2_TRACE -DG2_ABLATE=10 > /dev/null 2>&1
echo "== no blend, no products"; PIPS_LIB_PATH=$R/build/libpips_g2tr10.so timeout 300 python tools/g2_trace.py 2>&1 | grep -v amdgpu.ids
