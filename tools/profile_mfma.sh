#!/bin/sh
# MFMA-pipe utilisation of the matrix-core kernels (SQ_VALU_MFMA_BUSY_CYCLES, its own rocprofv3 --pmc pass: counters and
# --stats traces are never combined) for the headline command, the config-3 leg and the split-bf16 mode, plus the stock
# library yardsticks (rocBLAS / MIOpen / hipBLASLt through torch) that DESIGN.md quotes.  Run on the GPU box:
#   sh tools/profile_mfma.sh r3      -> gpurun_out/<tag>_pmc_mfma_util_{exact,config3,split}.txt, <tag>_probe_*.txt
TAG=${1:-r4}
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd /tmp
HEAD="python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-stage-profile --no-extras"
rm -rf /tmp/pm_*
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d /tmp/pm_head -o p -- $HEAD > /dev/null 2>&1
python $R/tools/mfma_util.py /tmp/pm_head > $R/gpurun_out/${TAG}_pmc_mfma_util_exact.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d /tmp/pm_c3 -o p -- python $R/bench.py --config 3 --steps 5 --warmup 2 --no-cpu-baseline --no-stage-profile --no-extras > /dev/null 2>&1
python $R/tools/mfma_util.py /tmp/pm_c3 > $R/gpurun_out/${TAG}_pmc_mfma_util_config3.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d /tmp/pm_split -o p -- $HEAD --matmul split > /dev/null 2>&1
python $R/tools/mfma_util.py /tmp/pm_split > $R/gpurun_out/${TAG}_pmc_mfma_util_split.txt 2>&1
python $R/tools/blas_probe.py > $R/gpurun_out/${TAG}_probe_blas_fp32.txt 2>&1
python $R/tools/conv_probe.py > $R/gpurun_out/${TAG}_probe_miopen_fp32.txt 2>&1
PIPS_PROBE_BLAS=1 python $R/tools/bf16_tile_probe.py 16384 > $R/gpurun_out/${TAG}_probe_bf16_gemm.txt 2>&1
tail -n 20 $R/gpurun_out/${TAG}_pmc_mfma_util_config3.txt
