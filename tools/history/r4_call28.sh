#!/bin/sh
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/pr_*
rocprofv3 --kernel-trace --stats -d /tmp/pr_enc -o p -- python $R/tools/encode_bench.py 8 368 496 > $O/c28_enc.log 2>&1
for f in $(find /tmp/pr_enc -name "*.db"); do python $R/tools/rocpd_summary.py $f $O/c28_enc_kernel_stats.txt > /dev/null; done
head -16 $O/c28_enc_kernel_stats.txt
