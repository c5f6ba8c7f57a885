#!/bin/sh
# round 4, GPU call 21: the four-wave bf16 GEMM kernels below one tile per CU (bf16 batches of 2 .. 7 clips per GPU)
R=$GRAFT_REPO_ROOT
cd $R
export PIPS_LIB_PATH=$R/pips_amd/libpips_hip_tune.so
{
for M in 4096 6144 8192 12288; do
for cfg in "PIPS_BF16_T4_MINPCT=100 PIPS_BF16_T4UP_MINPCT=100" "PIPS_BF16_T4_MINPCT=25 PIPS_BF16_T4UP_MINPCT=100" "PIPS_BF16_T4_MINPCT=100 PIPS_BF16_T4UP_MINPCT=25" "PIPS_BF16_T4_MINPCT=25 PIPS_BF16_T4UP_MINPCT=25"; do
  env $cfg timeout 200 python tools/mixer_bench.py $M bf16 2>/dev/null | sed "s/^/[$cfg] /"
done; done
} > gpurun_out/r4_call21_t4_small.log 2>&1
cat gpurun_out/r4_call21_t4_small.log
