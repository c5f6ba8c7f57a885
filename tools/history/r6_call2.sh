#!/bin/sh
# round 6, call 2: GPU suite on the hazard-guarded assembly (bitwise tests of the generated kernels), headline + config 3 timing
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests -m gpu -q > $O/r6c2_pytest_gpu.log 2>&1
echo "tests rc=$?"; tail -6 $O/r6c2_pytest_gpu.log
timeout 600 python bench.py --no-cpu-baseline --no-extras > $O/r6c2_bench_head.json 2> $O/r6c2_bench_head.err; tail -c 1200 $O/r6c2_bench_head.json
timeout 600 python bench.py --leg config3 2>/dev/null | tail -c 900
timeout 300 python tools/gather_c4.py 2>&1 | grep -v amdgpu.ids
