#!/bin/sh
# round 5, call 13: per-kernel times of the config-3 leg with the fp32 and the bf16 residual stream
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for st in f32 bf16; do
  rm -rf /tmp/pr_$st
  rocprofv3 --kernel-trace --stats -d /tmp/pr_$st -o p -- python $R/bench.py --leg config3 --mixer-stream $st > $O/r5c13_$st.log 2>&1
  for f in $(find /tmp/pr_$st -name "*.db"); do python $R/tools/rocpd_summary.py $f $O/r5c13_config3_kernel_stats_$st.txt > /dev/null; done
  echo "== stream $st"; head -8 $O/r5c13_config3_kernel_stats_$st.txt | cut -c1-70,108-150
done
cd $R
timeout 300 python -m pytest tests -m gpu -x -q -k "gemm_bf16_residual_stream" 2>&1 | tail -2
