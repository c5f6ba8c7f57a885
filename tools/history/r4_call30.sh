#!/bin/sh
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -q -m gpu > $O/c30_pytest_gpu.log 2>&1
echo "rc=$?" >> $O/c30_pytest_gpu.log
tail -8 $O/c30_pytest_gpu.log
