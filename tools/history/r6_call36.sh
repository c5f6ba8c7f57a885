#!/bin/sh
# round 6, call 36: MFMA rate and clock against the operand registers and the operand data (tools/mfma_operands.hip)
mkdir -p gpurun_out
./tools/mfma_operands > gpurun_out/r6_probe_mfma_operands.txt 2>&1; cat gpurun_out/r6_probe_mfma_operands.txt
