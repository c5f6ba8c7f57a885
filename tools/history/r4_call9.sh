#!/bin/sh
R=$GRAFT_REPO_ROOT
cd $R
{
for rep in 1 2; do
PIPS_LIB_PATH=$R/pips_amd/libpips_hip_tune.so PIPS_BF16_T4_DB=0 timeout 300 python tools/t4_check.py 2>&1 | grep "^\["
PIPS_LIB_PATH=$R/pips_amd/libpips_hip_tune.so PIPS_BF16_T4_DB=1 timeout 300 python tools/t4_check.py 2>&1 | grep "^\["
for gap in 2 10 14; do PIPS_LIB_PATH=$R/build/libpips_t4gap$gap.so PIPS_BF16_T4_DB=1 timeout 300 python tools/t4_check.py 2>&1 | grep "^\["; done
done
} > gpurun_out/r4_call9_t4_db.log 2>&1
cat gpurun_out/r4_call9_t4_db.log
