#!/bin/sh
# round 6, call 1: the GPU suite on the one-rounding-contract build (+ the RCCL one-rank test), the gathers' starting point,
# FETCH/WRITE counters of the config-3 leg (none existed for round 5), kernel stats of the config-3 leg
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests -m gpu -q > $O/r6c1_pytest_gpu.log 2>&1
echo "tests rc=$?"; tail -15 $O/r6c1_pytest_gpu.log
timeout 300 python tools/gather_c4.py > $O/r6c1_gather_c4.txt 2>&1; cat $O/r6c1_gather_c4.txt
export TMPDIR=/tmp; cd /tmp
C3="python $R/bench.py --leg config3"
rm -rf /tmp/pr_*
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pr_kt_c3 -o p -- $C3 > $O/r6c1_c3.log 2>&1
for f in $(find /tmp/pr_kt_c3 -name "*.db"); do python $R/tools/rocpd_summary.py $f $O/r6c1_config3_kernel_stats.txt > /dev/null; done
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pr_f_c3 -o p -- $C3 > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pr_w_c3 -o p -- $C3 > /dev/null 2>&1
python $R/tools/pmc_to_json.py $O/r6c1_pmc_traffic_config3.json /tmp/pr_f_c3 /tmp/pr_w_c3
head -30 $O/r6c1_config3_kernel_stats.txt
tail -c 1500 $O/r6c1_c3.log
