#!/bin/sh
# PMC counters of the two four-wave bf16 GEMM kernels (LDS bank conflicts, instruction mix, MFMA busy): one rocprofv3 --pmc pass per
# counter group (never combined with --stats).  usage (GPU box): sh tools/t4_pmc.sh [outfile]
OUT=${1:-gpurun_out/t4_pmc.txt}
GROUPS="SQ_LDS_BANK_CONFLICT:SQ_LDS_IDX_ACTIVE:SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU:SQ_INSTS_LDS:SQ_INSTS_VMEM:SQ_INSTS_SALU SQ_ACTIVE_INST_LDS:SQ_WAIT_INST_LDS:SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES:SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"
cd /tmp && export TMPDIR=/tmp
: > $GRAFT_REPO_ROOT/$OUT
for c in $GROUPS; do
  rm -rf /tmp/pmc_run
  rocprofv3 --kernel-trace --pmc $(echo $c | tr ':' ' ') -d /tmp/pmc_run -o p -- python $GRAFT_REPO_ROOT/tools/bf16_tile_probe.py 16384 > /tmp/pmc_log.txt 2>&1 || tail -3 /tmp/pmc_log.txt >> $GRAFT_REPO_ROOT/$OUT
  python $GRAFT_REPO_ROOT/tools/pmc_dump.py /tmp/pmc_run gemm_bf16_t4 >> $GRAFT_REPO_ROOT/$OUT 2>&1
done
cat $GRAFT_REPO_ROOT/$OUT
