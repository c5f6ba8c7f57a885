#!/bin/sh
# round 4, GPU call 4: fp32 mixer GEMM tile / swizzle A/B at config-4 size (M = 131072)
R=$GRAFT_REPO_ROOT
cd $R
export PIPS_LIB_PATH=$R/pips_amd/libpips_hip_tune.so
export PIPS_TOKEN_F32_MFMA=0
for cfg in "X=0" "PIPS_GEMM_TILE=10" "PIPS_GEMM_SWZ=1" "PIPS_GEMM_TILE=10 PIPS_GEMM_SWZ=1" "PIPS_GEMM_TILE=4" "X=0" "PIPS_GEMM_TILE=10"; do
  env $cfg python tools/mixer_bench.py 131072 2>/dev/null | sed "s/^/[$cfg] /"
done > gpurun_out/r4_call4_f32_tiles.log 2>&1
cat gpurun_out/r4_call4_f32_tiles.log
