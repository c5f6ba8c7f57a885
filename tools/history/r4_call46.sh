#!/bin/sh
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
LIBT=$R/pips_amd/libpips_hip_tune.so
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_forward_gpu.py -x -q -m gpu -k "gemm or mixer or golden or config2" > $O/c46_tests.log 2>&1
echo "rc=$?" >> $O/c46_tests.log; tail -4 $O/c46_tests.log | cut -c1-200
{
for v in 0 1 0 1 0 1; do
  PIPS_F32_T4_E=$v timeout 300 python bench.py --lib $LIBT --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('PIPS_F32_T4_E=$v headline', d['ms_per_step'], {k: round(v['ms']*1e3,2) for k,v in d['roofline']['all'].items()})"
done
} > $O/c46_e.txt 2>&1
cat $O/c46_e.txt
