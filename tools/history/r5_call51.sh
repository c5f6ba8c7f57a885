#!/bin/sh
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
for v in ab128 ab129; do
  echo "== $v" | tee -a $O/r5c48_ablation.txt
  PIPS_LIB_PATH=$R/build/libpips_$v.so timeout 300 python -u tools/gather_c4.py 2>&1 | grep -v amdgpu | grep "bf16 mode" | tee -a $O/r5c48_ablation.txt
done
