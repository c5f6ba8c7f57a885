#!/bin/sh
# round 4, GPU call 2: parity of the one-launch-per-layer mixer + A/B against the two-GEMM route
R=$GRAFT_REPO_ROOT
cd $R
python -m pytest tests/test_kernels_gpu.py tests/test_config3_gpu.py -m gpu -x -q -s -k "fused or mixer_bf16 or config3" > gpurun_out/r4_call2_tests.log 2>&1
tail -3 gpurun_out/r4_call2_tests.log
export PIPS_LIB_PATH=$R/pips_amd/libpips_hip_tune.so
PIPS_MIXER_LAYER=0 python tools/mixer_layer_ab.py 16384 > gpurun_out/r4_call2_mixer_ab.log 2>&1
PIPS_MIXER_LAYER=0 python tools/mixer_layer_ab.py 32768 >> gpurun_out/r4_call2_mixer_ab.log 2>&1
cat gpurun_out/r4_call2_mixer_ab.log
sh tools/ab_c3.sh PIPS_MIXER_LAYER 0 1 > gpurun_out/r4_call2_c3_ab.log 2>&1
cat gpurun_out/r4_call2_c3_ab.log
