#!/bin/sh
# round 6, call 10: token_mix_mfma_kernel with the GELU's min|x| as one v_med3 (a), and track.hip without IEEE-mode canonicalisation (b)
R=$GRAFT_REPO_ROOT; cd $R
sh tools/build_variant.sh tt track -DPIPS_TOKEN_TRACE > /dev/null 2>&1
echo "== (a) med3"; PIPS_LIB_PATH=$R/build/libpips_tt.so timeout 300 python tools/token_trace_bf16.py 2>&1 | grep "channel slots\|start ->"
sh tools/build_variant.sh ttb track -DPIPS_TOKEN_TRACE -fno-honor-nans -mno-amdgpu-ieee > /dev/null 2>&1
echo "== (b) med3 + no IEEE mode"; PIPS_LIB_PATH=$R/build/libpips_ttb.so timeout 300 python tools/token_trace_bf16.py 2>&1 | grep "channel slots\|start ->"
sh tools/build_variant.sh nb track -fno-honor-nans -mno-amdgpu-ieee > /dev/null 2>&1
for i in 1 2; do
  echo "product"; timeout 300 python bench.py --leg config3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1])['config3']; print(d['weak']['ms_per_step'])"
  echo "(b)"; timeout 300 python bench.py --lib $R/build/libpips_nb.so --leg config3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1])['config3']; print(d['weak']['ms_per_step'])"
done
