#!/bin/sh
# The round's rocprofv3 evidence (run on the GPU box through gpurun): kernel-trace summaries of the headline command and
# of the config-3 / config-4 legs, and FETCH_SIZE / WRITE_SIZE counter passes of the same two commands (separate runs: counters and
# --stats traces are never combined).  Outputs under gpurun_out/prof_<tag>_*; copy what is to be judged into profiles/.
# usage: sh tools/profile_round.sh r2
TAG=${1:-r4}
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd /tmp
HEAD="python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-stage-profile --no-extras"
C4="python $R/bench.py --leg config4"
C3="python $R/bench.py --leg config3"
# (the counter passes of the config-3 leg: rocprofv3 --pmc dies with SIGSEGV on the leg's ~20 k dispatches -- r6_call11 -- so they wrap the
#  short form of the same forward: 3 + 1 steps of B = 8)
C3P="python $R/bench.py --config 3 --steps 3 --warmup 1 --no-cpu-baseline --no-stage-profile --no-extras"
rm -rf /tmp/pr_*
rocprofv3 --kernel-trace --stats -d /tmp/pr_kt_head -o p -- $HEAD > $R/gpurun_out/prof_${TAG}_head.log 2>&1
rocprofv3 --kernel-trace --stats -d /tmp/pr_kt_c4 -o p -- $C4 > $R/gpurun_out/prof_${TAG}_c4.log 2>&1
rocprofv3 --kernel-trace --stats -d /tmp/pr_kt_c3 -o p -- $C3 > $R/gpurun_out/prof_${TAG}_c3.log 2>&1
for f in $(find /tmp/pr_kt_c3 -name "*.db"); do python $R/tools/rocpd_summary.py $f $R/gpurun_out/prof_${TAG}_config3_kernel_stats.txt > /dev/null; done
for f in $(find /tmp/pr_kt_head -name "*.db"); do python $R/tools/rocpd_summary.py $f $R/gpurun_out/prof_${TAG}_kernel_stats.txt > /dev/null; done
for f in $(find /tmp/pr_kt_c4 -name "*.db"); do python $R/tools/rocpd_summary.py $f $R/gpurun_out/prof_${TAG}_config4_kernel_stats.txt > /dev/null; done
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pr_f_head -o p -- $HEAD > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pr_w_head -o p -- $HEAD > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pr_f_c4 -o p -- $C4 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pr_w_c4 -o p -- $C4 > /dev/null 2>&1
python $R/tools/pmc_to_json.py $R/gpurun_out/prof_${TAG}_pmc_traffic.json /tmp/pr_f_head /tmp/pr_w_head /tmp/pr_f_c4 /tmp/pr_w_c4
# BASELINE configs[2] leg (bf16, B = 8 per GPU): its own file -- several kernel names also occur in the headline command
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pr_f_c3 -o p -- $C3P > $R/gpurun_out/prof_${TAG}_pmc_f_c3.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pr_w_c3 -o p -- $C3P > $R/gpurun_out/prof_${TAG}_pmc_w_c3.log 2>&1
python $R/tools/pmc_to_json.py $R/gpurun_out/prof_${TAG}_pmc_traffic_config3.json /tmp/pr_f_c3 /tmp/pr_w_c3
head -12 $R/gpurun_out/prof_${TAG}_config4_kernel_stats.txt
