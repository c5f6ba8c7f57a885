#!/bin/sh
# round 6, call 13: what clock does token_mix_mfma_kernel run at?  (s_memtime against the 100 MHz s_memrealtime, per wave)
mkdir -p gpurun_out
PIPS_LIB_PATH=build/libpips_tt.so python tools/token_trace_bf16.py > gpurun_out/r6c13_token_trace.txt 2>&1
echo "rc=$?"; cat gpurun_out/r6c13_token_trace.txt
