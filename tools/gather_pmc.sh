#!/bin/sh
# PMC counters of the tiled gather at config-4 geometry: one rocprofv3 --pmc pass per counter group
# usage (GPU box): sh tools/gather_pmc.sh [outfile] [fp32|bf16] [counter groups...]
OUT=${1:-gpurun_out/gather_pmc.txt}
MODE=${2:-fp32}
shift; shift
GROUPS="$@"
[ -z "$GROUPS" ] && GROUPS="FETCH_SIZE WRITE_SIZE SQ_LDS_BANK_CONFLICT:SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU:SQ_INSTS_LDS:SQ_INSTS_SALU:SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU:SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES:SQ_WAVE_CYCLES:SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
KERN=gather_tiled
[ "$MODE" = "bf16" ] && KERN=gather_mfma
cd /tmp && export TMPDIR=/tmp
: > $GRAFT_REPO_ROOT/$OUT
for c in $GROUPS; do
  rm -rf /tmp/pmc_run
  rocprofv3 --kernel-trace --pmc $(echo $c | tr ':' ' ') -d /tmp/pmc_run -o p -- python $GRAFT_REPO_ROOT/tools/gather_once.py 2 $MODE > /tmp/pmc_log.txt 2>&1 || tail -3 /tmp/pmc_log.txt >> $GRAFT_REPO_ROOT/$OUT
  python $GRAFT_REPO_ROOT/tools/pmc_dump.py /tmp/pmc_run $KERN >> $GRAFT_REPO_ROOT/$OUT 2>&1
done
cat $GRAFT_REPO_ROOT/$OUT
