#!/bin/sh
# round 5, call 12: the bf16 residual stream -- kernel tests, config-3 parity with both stream types, config-3 leg A/B
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q -s -k "residual_stream or test_gemm_bf16 or mixer_bf16" > $O/r5c12_tests.log 2>&1
echo "tests rc=$?"; grep -v "^$" $O/r5c12_tests.log | tail -14
for i in 1 2; do
for st in f32 bf16; do
  timeout 300 python bench.py --leg config3 --mixer-stream $st 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])['config3']
print('config3 leg, stream $st: weak %.3f ms  strong %.2f ms' % (d['weak']['ms_per_step'], d['strong']['ms_per_step']))"
done
done > $O/r5c12_ab.txt 2>&1
cat $O/r5c12_ab.txt
