#!/bin/sh
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
{
echo "GM_ABLATE bits: 1 no map loads, 2 no products / scatter, 4 no stores, 8 no feature loads, 16 items only"
timeout 100 python tools/gm_time.py 2>&1 | grep -v amdgpu.ids
for k in 16 15 14 13 11 1 2 4 3; do
  PIPS_LIB_PATH=$R/build/libpips_ab$k.so timeout 100 python tools/gm_time.py 2>&1 | grep -v amdgpu.ids | tail -1
done
} > $O/r5c9_ablate.txt 2>&1
cat $O/r5c9_ablate.txt
