#!/bin/sh
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/pr_*
rocprofv3 --kernel-trace -d /tmp/pr_gap -o p -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-stage-profile --no-extras > $O/c41.log 2>&1
for f in $(find /tmp/pr_gap -name "*.db"); do python $R/tools/rocpd_gaps.py $f > $O/c41_gaps.txt 2>&1; done
cat $O/c41_gaps.txt
