#!/bin/sh
# vector-memory instruction counts per kernel of the config-3 leg and of the headline (SQ_INSTS_VMEM, its own rocprofv3
# --pmc pass) next to the kernels' durations: a kernel whose (instructions per CU) x ~30 clocks approaches its duration is bound by
# the rate of its vector-memory instructions, not by bytes.   usage (GPU box): sh tools/vmem_count.sh
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd /tmp && rm -rf /tmp/vm_*
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM -d /tmp/vm_c3 -o p -- python $R/bench.py --config 3 --steps 5 --warmup 2 --no-cpu-baseline --no-stage-profile --no-extras > /dev/null 2>&1
python $R/tools/vmem_count.py /tmp/vm_c3 > $R/gpurun_out/vmem_count_config3.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM -d /tmp/vm_h -o p -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-stage-profile --no-extras > /dev/null 2>&1
python $R/tools/vmem_count.py /tmp/vm_h > $R/gpurun_out/vmem_count_headline.txt 2>&1
cat $R/gpurun_out/vmem_count_config3.txt
