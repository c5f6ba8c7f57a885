#!/bin/sh
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
LIBT=$R/pips_amd/libpips_hip_tune.so
for v in 0 1; do for m in 2048 4096; do PIPS_F32_T4_V=$v PIPS_LIB_PATH=$LIBT timeout 100 python tools/mixer_digest.py $m 2>&1 | grep digest; done; done
{
for rep in 1 2 3; do for v in 0 1; do
  echo "PIPS_F32_T4_V=$v"; PIPS_F32_T4_V=$v PIPS_LIB_PATH=$LIBT timeout 100 python tools/f32_t4_kscan.py 2048 2>&1 | grep "N=2048"
  PIPS_F32_T4_V=$v PIPS_LIB_PATH=$LIBT timeout 200 python tools/mixer_bench.py 2048 2>&1 | grep mixer
done; done
} > $O/c44_v.txt 2>&1
cat $O/c44_v.txt
