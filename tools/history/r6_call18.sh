#!/bin/sh
# round 6, call 18: the same write-through bit in the exact-fp32 mixer (headline shape, M = 2048): token_mix_kernel's stores, the fp32 GEMMs' stores, both
export PIPS_AB_M=2048 PIPS_AB_BF16=0
sh tools/tm_store_ab.sh tm32 f32sc1 f32all
mv gpurun_out/r6_probe_store_policy.txt gpurun_out/r6_probe_store_policy_fp32.txt
