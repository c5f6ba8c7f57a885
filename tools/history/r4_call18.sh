#!/bin/sh
# round 4, GPU call 18: K-split 4 for small-M fp32 GEMMs (late hops of chained tracking); config-4 test with both matrix modes
R=$GRAFT_REPO_ROOT
cd $R
export PIPS_LIB_PATH=$R/pips_amd/libpips_hip_tune.so
{
for M in 256 512 1024 1536 2048; do for t in -1 9 -1 9; do PIPS_GEMM_TILE=$t timeout 200 python tools/mixer_bench.py $M 2>/dev/null | sed "s/^/[TILE=$t] /"; done; done
for M in 512 1024; do for t in 5 9; do PIPS_GEMM_TILE_DOWN=$t timeout 200 python tools/mixer_bench.py $M 2>/dev/null | sed "s/^/[TILE_DOWN=$t] /"; done; done
} > gpurun_out/r4_call18_smallM.log 2>&1
cat gpurun_out/r4_call18_smallM.log
unset PIPS_LIB_PATH
timeout 900 python -m pytest tests/test_config45_gpu.py -m gpu -x -q -s -k "config4_end_to_end" > gpurun_out/r4_call18_c4test.log 2>&1
grep -v "^\[" gpurun_out/r4_call18_c4test.log | tail -6
