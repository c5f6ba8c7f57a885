#!/bin/sh
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
LIBT=$R/pips_amd/libpips_hip_tune.so
timeout 300 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm_f32 or gemm_identity" > $O/c25_tests.log 2>&1
echo "gemm tests rc=$?" >> $O/c25_tests.log
tail -3 $O/c25_tests.log
{
for rep in 1 2; do
for v in 0 1; do
  echo "PIPS_F32_T4_XCD=$v"; PIPS_F32_T4_XCD=$v PIPS_LIB_PATH=$LIBT timeout 200 python tools/f32_t4_kscan.py 2048 2>&1 | grep "M="
  PIPS_F32_T4_XCD=$v PIPS_LIB_PATH=$LIBT timeout 200 python tools/mixer_bench.py 2048 2>&1 | grep mixer
  PIPS_F32_T4_XCD=$v PIPS_LIB_PATH=$LIBT timeout 200 python tools/mixer_bench.py 2048 2>&1 | grep mixer
  PIPS_F32_T4_XCD=$v PIPS_LIB_PATH=$LIBT timeout 200 python tools/mixer_bench.py 16384 2>&1 | grep mixer
done; done
} > $O/c25_xcd.txt 2>&1
cat $O/c25_xcd.txt
