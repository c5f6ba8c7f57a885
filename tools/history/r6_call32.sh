#!/bin/sh
# round 6, call 32: the GPU suite on the final tree (two encoder cases added since the evidence run)
mkdir -p gpurun_out
python -m pytest tests -m gpu -q > gpurun_out/r6_pytest_gpu.log 2>&1; tail -3 gpurun_out/r6_pytest_gpu.log
