#!/bin/sh
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
: > $O/r5c41_ab.txt
for rep in 1 2 3; do for v in base seg; do
  echo "== $v" | tee -a $O/r5c41_ab.txt
  PIPS_LIB_PATH=$R/build/libpips_$v.so timeout 300 python -u tools/gather_c4.py 2>&1 | grep -v amdgpu | grep "bf16 mode" | tee -a $O/r5c41_ab.txt
done; done
PIPS_LIB_PATH=$R/build/libpips_base.so timeout 300 python tools/gather_dump.py /tmp/x_base.pt 2>&1 | grep -v amdgpu | tail -1
PIPS_LIB_PATH=$R/build/libpips_seg.so timeout 300 python tools/gather_dump.py /tmp/x_seg.pt 2>&1 | grep -v amdgpu | tail -1
python tools/gather_dump.py --compare /tmp/x_base.pt /tmp/x_seg.pt | head -2 | tee -a $O/r5c41_ab.txt
