#!/bin/sh
# round 6, call 38: the clock inside the two generated bf16 GEMMs (stamps around the statement), a wave's time against the launch
mkdir -p gpurun_out
PIPS_LIB_PATH=build/libpips_t4clk.so python tools/t4_clock.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r6_probe_t4_clock.txt; cat gpurun_out/r6_probe_t4_clock.txt
