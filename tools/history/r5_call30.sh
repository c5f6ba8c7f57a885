#!/bin/sh
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
: > $O/r5c30_wave_ablation.txt
for v in wave wab1 wab2 wab4 wab6 wab14 wab15; do
  echo "== $v" | tee -a $O/r5c30_wave_ablation.txt
  PIPS_LIB_PATH=$R/build/libpips_$v.so timeout 300 python -u tools/gather_c4.py 2>&1 | grep -v amdgpu | grep "bf16 mode" | tee -a $O/r5c30_wave_ablation.txt
done
