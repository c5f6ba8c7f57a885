#!/bin/sh
# same-box A/B of the config-3 leg over one tuning variable of the tuning library (pips_amd/libpips_hip_tune.so)
# usage: sh tools/ab_c3.sh PIPS_TOKEN_PP 1 2     (two rounds of each value, interleaved)
VAR=$1; A=$2; B=$3
R=$GRAFT_REPO_ROOT
LIBT=$R/pips_amd/libpips_hip_tune.so
for round in 1 2; do
  for v in $A $B; do
    env $VAR=$v python $R/bench.py --lib $LIBT --leg config3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])['config3']
print('$VAR=$v weak %.2f ms  strong %.1f ms' % (d['weak']['ms_per_step'], d['strong']['ms_per_step']))"
  done
done
