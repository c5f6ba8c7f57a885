#!/bin/sh
# round 6, call 39: the same stamps on the up-projection without its staging + fragment reads, and additionally without the GELU arithmetic
mkdir -p gpurun_out
for v in t4clk t4clkab t4clkab2; do echo "== $v"; PIPS_LIB_PATH=build/libpips_$v.so python tools/t4_clock.py 2>&1 | grep "after 300" | grep gelu; done > gpurun_out/r6_probe_t4_clock_ablated.txt; cat gpurun_out/r6_probe_t4_clock_ablated.txt
