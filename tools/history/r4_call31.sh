#!/bin/sh
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
{
for w in idle gemm2048 gemm16384 conv64; do timeout 60 python tools/clock_probe.py $w 2>&1 | grep -v amdgpu.ids; done
} > $O/c31_clock.txt 2>&1
cat $O/c31_clock.txt
