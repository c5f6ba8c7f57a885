#!/bin/sh
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "test_gemm_f32 or (test_conv_nhwc and not bf16)" > $O/c40_tests.log 2>&1
echo "rc=$?" >> $O/c40_tests.log
tail -12 $O/c40_tests.log | cut -c1-300
