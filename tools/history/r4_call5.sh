#!/bin/sh
# round 4, GPU call 5: XCD-aware tile order of the fp32 GEMM by problem size
R=$GRAFT_REPO_ROOT
cd $R
export PIPS_LIB_PATH=$R/pips_amd/libpips_hip_tune.so
for M in 4096 8192 16384 32768 65536; do
for cfg in "PIPS_GEMM_SWZ=0" "PIPS_GEMM_SWZ=1" "PIPS_GEMM_SWZ=0" "PIPS_GEMM_SWZ=1" "PIPS_GEMM_TILE=10 PIPS_GEMM_SWZ=1"; do
  env $cfg python tools/mixer_bench.py $M 2>/dev/null | sed "s/^/[$cfg] /"
done; done > gpurun_out/r4_call5_swz.log 2>&1
cat gpurun_out/r4_call5_swz.log
