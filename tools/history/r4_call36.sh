#!/bin/sh
# split-bf16 four-wave assembly GEMM: parity, then same-box A/B (hook PIPS_X3_T4 of the tuning library)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "split_bf16" > $O/c36_tests.log 2>&1
echo "x3 tests rc=$?" >> $O/c36_tests.log
tail -15 $O/c36_tests.log
if grep -q "failed\|rc=124\|error" $O/c36_tests.log; then exit 1; fi
LIBT=$R/pips_amd/libpips_hip_tune.so
{
for r in 1 2; do for v in 0 1; do
  echo "PIPS_X3_T4=$v"
  PIPS_LIB_PATH=$LIBT PIPS_X3_T4=$v timeout 200 python tools/mixer_bench.py 2048 x3 2>&1 | grep mixer
  PIPS_LIB_PATH=$LIBT PIPS_X3_T4=$v timeout 200 python tools/mixer_bench.py 16384 x3 2>&1 | grep mixer
done; done
} > $O/c36_ab.txt 2>&1
cat $O/c36_ab.txt
timeout 600 python -m pytest tests/test_forward_gpu.py -x -q -m gpu -k "split or x3 or golden" > $O/c36_fwd.log 2>&1
tail -3 $O/c36_fwd.log
for v in 0 1 0 1; do
  PIPS_X3_T4=$v timeout 300 python bench.py --lib $LIBT --matmul split --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('PIPS_X3_T4=$v split headline', d['ms_per_step'])" >> $O/c36_ab.txt 2>&1
done
tail -4 $O/c36_ab.txt
