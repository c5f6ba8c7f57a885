#!/bin/sh
# round 6, call 17: clock and package power (rocm-smi) under the bf16 mixer pass, against the fp32 GEMM loop of round 4's probe
mkdir -p gpurun_out
O=gpurun_out/r6_probe_clock_power_bf16.txt
echo "# shader clock and package power under sustained kernels (tools/clock_probe.py: rocm-smi sampled every 0.4 s while one workload loops for 3 s)" > $O
python tools/clock_probe.py mixerbf16_16384 2>/dev/null >> $O
python tools/clock_probe.py gemm16384 2>/dev/null >> $O
python tools/clock_probe.py mixerbf16_2048 2>/dev/null >> $O
cat $O
