#!/bin/sh
# round 4, GPU call 3: fp32 token mix on v_mfma_f32_4x4x1 -- parity + A/B
R=$GRAFT_REPO_ROOT
cd $R
python -m pytest tests/test_kernels_gpu.py tests/test_forward_gpu.py -m gpu -x -q -k "mixer or golden or config2" > gpurun_out/r4_call3_tests.log 2>&1
tail -3 gpurun_out/r4_call3_tests.log
export PIPS_LIB_PATH=$R/pips_amd/libpips_hip_tune.so
for M in 2048 131072; do for v in 0 1 0 1; do PIPS_TOKEN_F32_MFMA=$v python tools/mixer_bench.py $M | sed "s/^/F32_MFMA=$v /"; done; done > gpurun_out/r4_call3_tokmix_ab.log 2>&1
cat gpurun_out/r4_call3_tokmix_ab.log
cd /tmp; export TMPDIR=/tmp
PIPS_TOKEN_F32_MFMA=1 rocprofv3 --kernel-trace --stats -d /tmp/pk1 -o p -- python $R/tools/mixer_bench.py 2048 > /dev/null 2>&1
for f in $(find /tmp/pk1 -name "*.db"); do python $R/tools/rocpd_summary.py $f $R/gpurun_out/r4_call3_mixer2048_stats.txt > /dev/null; done
PIPS_TOKEN_F32_MFMA=1 rocprofv3 --kernel-trace --stats -d /tmp/pk2 -o p -- python $R/tools/mixer_bench.py 131072 > /dev/null 2>&1
for f in $(find /tmp/pk2 -name "*.db"); do python $R/tools/rocpd_summary.py $f $R/gpurun_out/r4_call3_mixer131072_stats.txt > /dev/null; done
head -8 $R/gpurun_out/r4_call3_mixer2048_stats.txt; head -8 $R/gpurun_out/r4_call3_mixer131072_stats.txt
