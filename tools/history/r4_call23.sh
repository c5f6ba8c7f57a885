#!/bin/sh
# fp32 four-wave assembly GEMMs: parity, then same-box A/B (hook PIPS_F32_T4 of the tuning library)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm_f32 or gemm_identity" > $O/c23_tests.log 2>&1
echo "gemm tests rc=$?" >> $O/c23_tests.log
tail -15 $O/c23_tests.log
if grep -q "failed\|rc=124\|error" $O/c23_tests.log; then exit 1; fi
LIBT=$R/pips_amd/libpips_hip_tune.so
{
for r in 1 2; do for v in 0 1; do
  echo "PIPS_F32_T4=$v"; PIPS_LIB_PATH=$LIBT PIPS_F32_T4=$v timeout 200 python tools/gemm_bench.py --gemm-only 2>&1 | grep gemm
  PIPS_LIB_PATH=$LIBT PIPS_F32_T4=$v timeout 200 python tools/mixer_bench.py 2048 2>&1 | grep mixer
done; done
} > $O/c23_ab.txt 2>&1
cat $O/c23_ab.txt
timeout 600 python -m pytest tests/test_forward_gpu.py -x -q -m gpu > $O/c23_fwd.log 2>&1
tail -3 $O/c23_fwd.log
for v in 0 1 0 1; do
  PIPS_F32_T4=$v timeout 300 python bench.py --lib $LIBT --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('PIPS_F32_T4=$v', d['ms_per_step'], d['roofline'])" >> $O/c23_ab.txt 2>&1
done
tail -4 $O/c23_ab.txt
