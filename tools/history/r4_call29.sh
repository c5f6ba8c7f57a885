#!/bin/sh
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
LIBT=$R/pips_amd/libpips_hip_tune.so
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "test_conv_nhwc and not bf16" > $O/c29_tests.log 2>&1
echo "conv tests rc=$?" >> $O/c29_tests.log
tail -3 $O/c29_tests.log
{
for r in 1 2 3; do for v in 0 1; do
  echo "PIPS_CONV_F32_T4_XCD=$v"; PIPS_LIB_PATH=$LIBT PIPS_CONV_F32_T4_XCD=$v timeout 200 python tools/encode_bench.py 8 368 496
done; done
} > $O/c29_ab.txt 2>&1
cat $O/c29_ab.txt
