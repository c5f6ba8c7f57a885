#!/bin/sh
# round 6: the evidence of the final code -- GPU suite, bench line, rocprofv3 summaries, PMC passes (copy what is to be judged into profiles/)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -q > $O/r6_pytest_gpu.log 2>&1
echo "tests rc=$?"; tail -3 $O/r6_pytest_gpu.log
timeout 1200 python bench.py > $O/r6_bench.json 2> $O/r6_bench.err
echo "bench rc=$?"; tail -c 400 $O/r6_bench.json
timeout 2400 sh tools/profile_round.sh r6 > $O/r6_profile.log 2>&1
echo "profile rc=$?"
timeout 900 sh tools/profile_mfma.sh r6 > $O/r6_profile_mfma.log 2>&1
echo "mfma rc=$?"
sh tools/gather_pmc.sh gpurun_out/r6_gather_pmc_counters_fp32.txt fp32 > /dev/null 2>&1
sh tools/gather_pmc.sh gpurun_out/r6_gather_pmc_counters_bf16.txt bf16 FETCH_SIZE WRITE_SIZE TCC_HIT_sum:TCC_MISS_sum:TCC_REQ_sum SQ_LDS_BANK_CONFLICT:SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU:SQ_INSTS_LDS:SQ_INSTS_SALU:SQ_INSTS_VMEM SQ_BUSY_CYCLES:SQ_WAVE_CYCLES:SQ_WAIT_INST_LDS SQ_WAIT_ANY:SQ_WAIT_INST_ANY:SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES:SQ_INSTS_MFMA GRBM_GUI_ACTIVE > /dev/null 2>&1
timeout 300 python tools/gather_c4.py > $O/r6_probe_gather_final.txt 2>&1
sh tools/build_variant.sh tt track -DPIPS_TOKEN_TRACE -fno-honor-nans -mno-amdgpu-ieee > /dev/null 2>&1
PIPS_LIB_PATH=$R/build/libpips_tt.so timeout 300 python tools/token_trace_bf16.py 2>&1 | grep -v "amdgpu.ids\|first block" > $O/r6_probe_token_mix_mfma_phases.txt
ls $O | grep "r6_\|prof_r6" | head -40
