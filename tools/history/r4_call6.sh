#!/bin/sh
# round 4, GPU call 6: XCD-aware tile order in the conv kernels (fp32 / bf16 / split), mixer row chunks at M = 131072
R=$GRAFT_REPO_ROOT
cd $R
export PIPS_LIB_PATH=$R/pips_amd/libpips_hip_tune.so
{
for v in 0 1 0 1; do PIPS_GEMM_SWZ=$v python tools/encode_bench.py 8 368 496 f32 2>/dev/null | sed "s/^/[GEMM_SWZ=$v] /"; done
for v in 0 1 0 1; do PIPS_GEMM_SWZ=$v python tools/encode_bench.py 32 720 1280 f32 2>/dev/null | sed "s/^/[GEMM_SWZ=$v] /"; done
for v in 0 1 0 1; do PIPS_BF16_SWZ=$v python tools/encode_bench.py 64 368 496 bf16 2>/dev/null | sed "s/^/[BF16_SWZ=$v] /"; done
for v in 0 1 0 1; do PIPS_BF16_SWZ=$v python tools/encode_bench.py 8 368 496 bf16 2>/dev/null | sed "s/^/[BF16_SWZ=$v] /"; done
for v in 0 1 0 1; do PIPS_X3_SWZ=$v python tools/encode_bench.py 8 368 496 split 2>/dev/null | sed "s/^/[X3_SWZ=$v] /"; done
for v in 0 65536 32768 0 65536; do PIPS_MIXER_CHUNK=$v python tools/mixer_bench.py 131072 2>/dev/null | sed "s/^/[MIXER_CHUNK=$v] /"; done
for v in 0 65536; do PIPS_MIXER_CHUNK=$v PIPS_GEMM_SWZ=1 python tools/mixer_bench.py 131072 2>/dev/null | sed "s/^/[MIXER_CHUNK=$v GEMM_SWZ=1] /"; done
} > gpurun_out/r4_call6_swz_conv.log 2>&1
cat gpurun_out/r4_call6_swz_conv.log
