#!/bin/sh
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
HEAD="python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-stage-profile --no-extras"
rm -rf /tmp/pr_*
rocprofv3 --kernel-trace --stats -d /tmp/pr_kt_head -o p -- $HEAD > $O/c26_head.log 2>&1
for f in $(find /tmp/pr_kt_head -name "*.db"); do python $R/tools/rocpd_summary.py $f $O/c26_kernel_stats.txt > /dev/null; done
head -30 $O/c26_kernel_stats.txt
tail -1 $O/c26_head.log | cut -c1-400
