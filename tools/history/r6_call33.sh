#!/bin/sh
# round 6, call 33: clock and package power with every SIMD issuing MFMAs back to back (tools/mfma_power.hip), rocm-smi sampled beside it
mkdir -p gpurun_out
O=gpurun_out/r6_probe_mfma_clock_power.txt
(while true; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Package Power" | sed 's/.*sclk clock level: [0-9S]*: (\([0-9]*\)Mhz).*/sclk \1 MHz/; s/.*Package Power (W): \([0-9.]*\).*/\1 W/' | tr '\n' ' '; echo; sleep 0.5; done) > /tmp/smi.log 2>&1 &
SMI=$!
./tools/mfma_power 4 > $O 2>&1
kill $SMI
echo "# rocm-smi every 0.5 s while the six loops above ran (4 s each):" >> $O
awk 'NF' /tmp/smi.log | awk 'NR%2==1' | head -40 >> $O
cat $O
