#!/bin/sh
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
{
for rep in 1 2; do
for L in $R/pips_amd/libpips_hip_tune.so $R/build/libpips_noflags.so $R/build/libpips_nostats.so $R/build/libpips_noepi.so; do
  echo "lib $(basename $L)"; PIPS_LIB_PATH=$L timeout 60 python tools/clock_probe.py conv64 2>&1 | grep "TF sustained"
done; done
} > $O/c32_ablate.txt 2>&1
cat $O/c32_ablate.txt
