#!/bin/sh
# evidence of the final gather_mfma_kernel: whole GPU suite, config-4 / config-3 legs under rocprofv3, PMC passes of the gather, bench line
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -m gpu -q > $O/r5_pytest_gpu.log 2>&1
echo "tests rc=$?"; tail -3 $O/r5_pytest_gpu.log
timeout 900 python bench.py > $O/r5_bench.json 2> $O/r5_bench.err
echo "bench rc=$?"; tail -c 300 $O/r5_bench.json
timeout 1500 sh tools/profile_round.sh r5 > $O/r5_profile.log 2>&1
echo "profile rc=$?"
sh tools/gather_pmc.sh gpurun_out/r5_gather_pmc_counters_bf16.txt bf16 FETCH_SIZE WRITE_SIZE TCC_HIT_sum:TCC_MISS_sum:TCC_REQ_sum SQ_LDS_BANK_CONFLICT:SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU:SQ_INSTS_LDS:SQ_INSTS_SALU:SQ_INSTS_VMEM SQ_BUSY_CYCLES:SQ_WAVE_CYCLES:SQ_WAIT_INST_LDS SQ_WAIT_ANY:SQ_WAIT_INST_ANY:SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES:SQ_INSTS_MFMA GRBM_GUI_ACTIVE > /dev/null 2>&1
cat $O/r5_gather_pmc_counters_bf16.txt
