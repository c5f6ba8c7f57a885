#!/bin/sh
# round 4, GPU call 10: whole GPU test suite + bench line of HEAD
R=$GRAFT_REPO_ROOT
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r4_call10_pytest_gpu.log 2>&1
tail -5 gpurun_out/r4_call10_pytest_gpu.log
timeout 900 python bench.py > gpurun_out/r4_call10_bench.json 2> gpurun_out/r4_call10_bench.err
tail -c 3000 gpurun_out/r4_call10_bench.json
