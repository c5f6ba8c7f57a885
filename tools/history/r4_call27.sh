#!/bin/sh
# fp32 four-wave assembly convolutions: parity, then same-box A/B (hook PIPS_CONV_F32_T4 of the tuning library)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "test_conv_nhwc and not bf16" > $O/c27_tests.log 2>&1
echo "conv tests rc=$?" >> $O/c27_tests.log
tail -25 $O/c27_tests.log
if grep -q "failed\|rc=124\|error" $O/c27_tests.log; then exit 1; fi
LIBT=$R/pips_amd/libpips_hip_tune.so
{
for r in 1 2; do for v in 0 1; do
  echo "PIPS_CONV_F32_T4=$v"; PIPS_LIB_PATH=$LIBT PIPS_CONV_F32_T4=$v timeout 200 python tools/encode_bench.py 8 368 496
done; done
} > $O/c27_ab.txt 2>&1
cat $O/c27_ab.txt
timeout 600 python -m pytest tests/test_forward_gpu.py -x -q -m gpu > $O/c27_fwd.log 2>&1
tail -3 $O/c27_fwd.log
for v in 0 1 0 1; do
  PIPS_CONV_F32_T4=$v timeout 300 python bench.py --lib $LIBT --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('PIPS_CONV_F32_T4=$v', d['ms_per_step'])" >> $O/c27_ab.txt 2>&1
done
tail -4 $O/c27_ab.txt
