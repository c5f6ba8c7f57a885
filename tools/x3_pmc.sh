#!/bin/sh
# PMC counters of the split-bf16 GEMM (gemm_x3_kernel) on the two channel-mix shapes at M = 2048 and M = 131072: one rocprofv3 --pmc
# pass per counter group and shape (never combined with --stats).  usage (GPU box): sh tools/x3_pmc.sh [outfile]
OUT=${1:-gpurun_out/x3_pmc.txt}
GROUPS="SQ_LDS_BANK_CONFLICT:SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU:SQ_INSTS_LDS:SQ_INSTS_VMEM:SQ_INSTS_SALU SQ_ACTIVE_INST_LDS:SQ_WAIT_INST_LDS:SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES:SQ_WAVE_CYCLES:SQ_INSTS_MFMA SQ_WAIT_ANY:SQ_WAIT_INST_ANY:SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
cd /tmp && export TMPDIR=/tmp
: > $GRAFT_REPO_ROOT/$OUT
for shape in "2048 2048 512 1 20" "2048 512 2048 2 20" "131072 2048 512 1 4" "131072 512 2048 2 4"; do
  echo "== M N K epi reps: $shape" >> $GRAFT_REPO_ROOT/$OUT
  for c in $GROUPS; do
    rm -rf /tmp/pmc_run
    rocprofv3 --kernel-trace --pmc $(echo $c | tr ':' ' ') -d /tmp/pmc_run -o p -- python $GRAFT_REPO_ROOT/tools/x3_one.py $shape > /tmp/pmc_log.txt 2>&1 || tail -3 /tmp/pmc_log.txt >> $GRAFT_REPO_ROOT/$OUT
    python $GRAFT_REPO_ROOT/tools/pmc_dump.py /tmp/pmc_run gemm_x3 >> $GRAFT_REPO_ROOT/$OUT 2>&1
  done
  rm -rf /tmp/pmc_run
  rocprofv3 --kernel-trace --stats -d /tmp/pmc_run -o p -- python $GRAFT_REPO_ROOT/tools/x3_one.py $shape > /tmp/pmc_log.txt 2>&1
  for f in $(find /tmp/pmc_run -name "*.db"); do python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $f /tmp/x3_stats.txt > /dev/null; grep gemm_x3 /tmp/x3_stats.txt | cut -c1-60,108-160 >> $GRAFT_REPO_ROOT/$OUT; done
done
cat $GRAFT_REPO_ROOT/$OUT
