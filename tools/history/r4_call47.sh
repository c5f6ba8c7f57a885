#!/bin/sh
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
{
for rep in 1 2; do for L in $R/pips_amd/libpips_hip_tune.so $R/build/libpips_trold.so; do
  echo "lib $(basename $L)"
  PIPS_LIB_PATH=$L timeout 200 python tools/mixer_bench.py 2048 2>&1 | grep mixer
  PIPS_LIB_PATH=$L timeout 200 python tools/mixer_bench.py 131072 2>&1 | grep mixer
done; done
} > $O/c47_trold.txt 2>&1
cat $O/c47_trold.txt
