#!/bin/sh
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
{
for i in 1 2 3; do
  timeout 300 python bench.py --no-extras --no-cpu-baseline --no-stage-profile --steps 100 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('exact  steps=100: %.3f ms per forward, %.4f M updates/s' % (d['ms_per_step'], d['value']/1e6))"
done
for i in 1 2 3; do
  timeout 300 python bench.py --matmul split --no-extras --no-cpu-baseline --no-stage-profile --steps 100 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('split  steps=100: %.3f ms per forward, %.4f M updates/s' % (d['ms_per_step'], d['value']/1e6))"
done
} > $O/c48_spread.txt 2>&1
cat $O/c48_spread.txt
