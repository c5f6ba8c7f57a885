#!/bin/sh
# round 4, GPU call 8: the four-wave 128x256 down-projection kernel -- parity, then timing against the 256x128 assembly kernel
R=$GRAFT_REPO_ROOT
cd $R
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_config3_gpu.py -m gpu -x -q -s -k "gemm_bf16 or mixer_bf16 or config3" > gpurun_out/r4_call8_tests.log 2>&1
tail -4 gpurun_out/r4_call8_tests.log
export PIPS_LIB_PATH=$R/pips_amd/libpips_hip_tune.so
{
for v in 0 1 0 1; do PIPS_BF16_T4=$v timeout 300 python tools/bf16_res_probe.py 2>/dev/null | sed "s/^/[T4=$v] /"; done
} > gpurun_out/r4_call8_t4.log 2>&1
cat gpurun_out/r4_call8_t4.log
timeout 600 sh tools/ab_c3.sh PIPS_BF16_T4 0 1 > gpurun_out/r4_call8_c3_ab.log 2>&1
cat gpurun_out/r4_call8_c3_ab.log
