#!/bin/sh
# eight-wave token-mix kernel: parity, then same-box A/B (hook PIPS_TOKEN_W8 of the tuning library)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_forward_gpu.py -x -q -m gpu -k "mixer or golden or config2 or token" > $O/c38_tests.log 2>&1
echo "tests rc=$?" >> $O/c38_tests.log
tail -6 $O/c38_tests.log
if grep -q "failed\|rc=124\|error" $O/c38_tests.log; then exit 1; fi
LIBT=$R/pips_amd/libpips_hip_tune.so
{
for r in 1 2; do for v in 0 1; do
  echo "PIPS_TOKEN_W8=$v"
  PIPS_LIB_PATH=$LIBT PIPS_TOKEN_W8=$v timeout 200 python tools/mixer_bench.py 2048 2>&1 | grep mixer
  PIPS_LIB_PATH=$LIBT PIPS_TOKEN_W8=$v timeout 200 python tools/mixer_bench.py 131072 2>&1 | grep mixer
done; done
for v in 0 1 0 1; do
  PIPS_TOKEN_W8=$v timeout 300 python bench.py --lib $LIBT --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('PIPS_TOKEN_W8=$v headline', d['ms_per_step'])"
done
} > $O/c38_ab.txt 2>&1
cat $O/c38_ab.txt
