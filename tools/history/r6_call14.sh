#!/bin/sh
# round 6, call 14: the clock again, after 3 / 30 / 300 passes; what rocm-smi says about the clock levels and the power cap
mkdir -p gpurun_out
(rocm-smi --showclocks --showpower --showmaxpower --showperflevel 2>&1 | head -40) > gpurun_out/r6c14_rocm_smi.txt
PIPS_LIB_PATH=build/libpips_tt.so python tools/token_trace_bf16.py > gpurun_out/r6c14_token_trace.txt 2>&1
echo "rc=$?"; cat gpurun_out/r6c14_token_trace.txt; cat gpurun_out/r6c14_rocm_smi.txt
(rocm-smi --showclocks --showpower 2>&1 | head -30) >> gpurun_out/r6c14_rocm_smi.txt
