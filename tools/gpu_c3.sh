#!/bin/sh
# quick config-3 loop on the GPU box: bf16/config3 tests, then the rocprofv3 kernel summary of the config-3 leg
# usage: sh tools/gpu_c3.sh <tag> [pytest -k expression]
TAG=${1:-x}
K=${2:-"bf16 or config3"}
R=$GRAFT_REPO_ROOT
python -m pytest $R/tests -m gpu -x -q -s -k "$K" > $R/gpurun_out/${TAG}_tests.log 2>&1; echo rc=$? >> $R/gpurun_out/${TAG}_tests.log
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pr_kt_c3
rocprofv3 --kernel-trace --stats -d /tmp/pr_kt_c3 -o p -- python $R/bench.py --leg config3 > $R/gpurun_out/${TAG}_c3.log 2>&1
for f in $(find /tmp/pr_kt_c3 -name "*.db"); do python $R/tools/rocpd_summary.py $f $R/gpurun_out/${TAG}_config3_kernel_stats.txt > /dev/null; done
python $R/bench.py --leg config3 > $R/gpurun_out/${TAG}_config3.json 2>/dev/null
tail -n 12 $R/gpurun_out/${TAG}_tests.log | cut -c1-250
head -24 $R/gpurun_out/${TAG}_config3_kernel_stats.txt | cut -c1-60,110-170
python -c "import json;d=json.load(open('$R/gpurun_out/${TAG}_config3.json'))['config3'];print('config3 ms', d['ms_per_step'])"
