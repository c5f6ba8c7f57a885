#!/bin/sh
# round 4, GPU call 12: the four-wave 256x256 up-projection kernel -- parity, then timing against gemm_bf16_gelu256_asm_kernel
R=$GRAFT_REPO_ROOT
cd $R
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_config3_gpu.py -m gpu -x -q -s -k "gemm_bf16 or mixer_bf16 or config3" > gpurun_out/r4_call12_tests.log 2>&1
tail -4 gpurun_out/r4_call12_tests.log
export PIPS_LIB_PATH=$R/pips_amd/libpips_hip_tune.so
{
for v in 0 1 0 1; do PIPS_BF16_T4UP=$v timeout 300 python tools/bf16_tile_probe.py 16384 2>/dev/null | sed "s/^/[T4UP=$v] /"; done
for v in 0 1; do PIPS_BF16_T4UP=$v timeout 300 python tools/bf16_tile_probe.py 32768 2>/dev/null | sed "s/^/[T4UP=$v] /"; done
} > gpurun_out/r4_call12_t4up.log 2>&1
cat gpurun_out/r4_call12_t4up.log
timeout 600 sh tools/ab_c3.sh PIPS_BF16_T4UP 0 1 > gpurun_out/r4_call12_c3_ab.log 2>&1
cat gpurun_out/r4_call12_c3_ab.log
