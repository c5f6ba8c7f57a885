#!/bin/sh
# round 4, GPU call 17: two waves per particle / degree-5 GELU in the bf16 token mix
R=$GRAFT_REPO_ROOT
cd $R
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_config3_gpu.py tests/test_forward_gpu.py -m gpu -x -q -k "mixer or config3 or bf16" > gpurun_out/r4_call17_tests.log 2>&1
tail -4 gpurun_out/r4_call17_tests.log
T=$R/pips_amd/libpips_hip_tune.so
{
for rep in 1 2; do
for cfg in "PIPS_TOKEN_WPP=1 PIPS_TOKEN_GELU5=0" "PIPS_TOKEN_WPP=1 PIPS_TOKEN_GELU5=1" "PIPS_TOKEN_WPP=2 PIPS_TOKEN_GELU5=0" "PIPS_TOKEN_WPP=2 PIPS_TOKEN_GELU5=1"; do
  env PIPS_LIB_PATH=$T $cfg timeout 300 python tools/mixer_bench.py 16384 bf16 2>/dev/null | sed "s/^/[$cfg] /"
done
env PIPS_LIB_PATH=$R/build/libpips_tokocc3.so PIPS_TOKEN_WPP=2 PIPS_TOKEN_GELU5=1 timeout 300 python tools/mixer_bench.py 16384 bf16 2>/dev/null | sed "s/^/[occ3 WPP=2 GELU5=1] /"
done
for cfg in "PIPS_TOKEN_WPP=1 PIPS_TOKEN_GELU5=0" "PIPS_TOKEN_WPP=2 PIPS_TOKEN_GELU5=1"; do
  env PIPS_LIB_PATH=$T $cfg timeout 300 python tools/mixer_bench.py 2048 bf16 2>/dev/null | sed "s/^/[$cfg] /"
done
} > gpurun_out/r4_call17_tokmix.log 2>&1
cat gpurun_out/r4_call17_tokmix.log
