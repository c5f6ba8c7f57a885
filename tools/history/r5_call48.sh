#!/bin/sh
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
: > $O/r5c48_ablation.txt
for v in trace_none ab1 ab2 ab8 ab32 ab33 ab34; do
  echo "== $v" | tee -a $O/r5c48_ablation.txt
  L=$R/build/libpips_$v.so; [ "$v" = trace_none ] && L=$R/pips_amd/libpips_hip.so
  PIPS_LIB_PATH=$L timeout 300 python -u tools/gather_c4.py 2>&1 | grep -v amdgpu | grep "bf16 mode" | tee -a $O/r5c48_ablation.txt
done
