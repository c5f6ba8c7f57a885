#!/bin/sh
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/pr_*
rocprofv3 --kernel-trace --stats -d /tmp/pr_split -o p -- python $R/bench.py --matmul split --steps 20 --warmup 3 --no-cpu-baseline --no-stage-profile --no-extras > $O/c37_split.log 2>&1
for f in $(find /tmp/pr_split -name "*.db"); do python $R/tools/rocpd_summary.py $f $O/c37_split_kernel_stats.txt > /dev/null; done
head -24 $O/c37_split_kernel_stats.txt
