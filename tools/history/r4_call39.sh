#!/bin/sh
# GELU moved from the up-projection's epilogue into the down-projection's operand staging: parity, bitwise A/B, timing A/B (hook PIPS_F32_T4_AGELU)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_forward_gpu.py -x -q -m gpu -k "mixer or golden or config2 or gemm_f32" > $O/c39_tests.log 2>&1
echo "tests rc=$?" >> $O/c39_tests.log
tail -6 $O/c39_tests.log
if grep -q "failed\|rc=124\|error" $O/c39_tests.log; then exit 1; fi
LIBT=$R/pips_amd/libpips_hip_tune.so
{
for v in 0 1; do for m in 2048 4096 16384; do
  echo "PIPS_F32_T4_AGELU=$v"; PIPS_LIB_PATH=$LIBT PIPS_F32_T4_AGELU=$v timeout 100 python tools/mixer_digest.py $m 2>&1 | grep digest
done; done
for r in 1 2; do for v in 0 1; do
  echo "PIPS_F32_T4_AGELU=$v"
  PIPS_LIB_PATH=$LIBT PIPS_F32_T4_AGELU=$v timeout 200 python tools/mixer_bench.py 2048 2>&1 | grep mixer
  PIPS_LIB_PATH=$LIBT PIPS_F32_T4_AGELU=$v timeout 200 python tools/mixer_bench.py 131072 2>&1 | grep mixer
done; done
for v in 0 1 0 1; do
  PIPS_F32_T4_AGELU=$v timeout 300 python bench.py --lib $LIBT --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('PIPS_F32_T4_AGELU=$v headline', d['ms_per_step'], d['roofline']['all'])"
done
} > $O/c39_ab.txt 2>&1
cat $O/c39_ab.txt | cut -c1-330
