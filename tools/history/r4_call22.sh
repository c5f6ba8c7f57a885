#!/bin/sh
# conv 96->96 four-wave kernel: parity, then same-box A/B (hook PIPS_CONV_C96_T4 of the tuning library)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "conv_nhwc_bf16" > $O/c22_tests.log 2>&1
echo "conv tests rc=$?" >> $O/c22_tests.log
tail -5 $O/c22_tests.log
if grep -q "failed\|rc=124" $O/c22_tests.log; then exit 1; fi
timeout 400 python -m pytest tests/test_config3_gpu.py tests/test_forward_gpu.py -x -q -m gpu -k "config3 or bf16" >> $O/c22_tests.log 2>&1
echo "fwd tests rc=$?" >> $O/c22_tests.log
tail -3 $O/c22_tests.log
LIBT=$R/pips_amd/libpips_hip_tune.so
{
for r in 1 2; do for v in 0 1; do
  echo "PIPS_CONV_C96_T4=$v"; PIPS_LIB_PATH=$LIBT PIPS_CONV_C96_T4=$v timeout 200 python tools/encode_bench.py 64 368 496 bf16
done; done
sh tools/ab_c3.sh PIPS_CONV_C96_T4 0 1
} > $O/c22_ab.txt 2>&1
cat $O/c22_ab.txt
cd /tmp && export TMPDIR=/tmp
PIPS_LIB_PATH=$LIBT timeout 300 rocprofv3 --kernel-trace --stats -d $O/c22_prof -o c22 -- python $R/tools/encode_bench.py 64 368 496 bf16 > /dev/null 2>&1
python - <<PY
import csv,glob
for f in glob.glob("$O/c22_prof/**/*kernel_stats.csv", recursive=True):
    rows=list(csv.DictReader(open(f)))
    for r in rows[:14]: print(r["Name"][:70], r["Calls"], r["AverageNs"], r["Percentage"])
PY
