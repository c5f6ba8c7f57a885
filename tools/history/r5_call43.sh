#!/bin/sh
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 300 python tools/gather_dump.py /tmp/x_prod.pt 2>&1 | grep -v amdgpu | tail -1
: > $O/r5c43_lookup.txt
for v in lk5 lk6; do
  PIPS_LIB_PATH=$R/build/libpips_$v.so timeout 300 python tools/gather_dump.py /tmp/x_$v.pt 2>&1 | grep -v amdgpu | tail -1
  echo "== $v" | tee -a $O/r5c43_lookup.txt
  python tools/gather_dump.py --compare /tmp/x_prod.pt /tmp/x_$v.pt | head -2 | tee -a $O/r5c43_lookup.txt
done
