#!/bin/sh
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
{
for rep in 1 2; do
for L in $R/pips_amd/libpips_hip_tune.so $R/build/libpips_f4nobar.so $R/build/libpips_f4nold.so $R/build/libpips_f4noldnost.so $R/build/libpips_f4noldnostnobar.so; do
  echo "lib $(basename $L)"; PIPS_LIB_PATH=$L timeout 100 python tools/f32_t4_kscan.py 2048 2>&1 | grep "M="
done; done
} > $O/c43_ablate.txt 2>&1
cat $O/c43_ablate.txt
