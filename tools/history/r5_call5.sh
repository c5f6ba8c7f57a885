#!/bin/sh
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 120 python tools/gm_debug2.py 0 2>&1 | grep -v amdgpu.ids | tail -40 > $O/r5c5_debug.txt
timeout 900 python -m pytest tests -m gpu -x -q -s -k "mfma or bf16_matrix or bf16_mode_end_to_end or mixer_input_build_tiled or config4_gather" > $O/r5c5_tests.log 2>&1
echo "tests rc=$?"; grep -v "^$" $O/r5c5_tests.log | tail -25
timeout 300 python -u tools/gather_c4.py > $O/r5c5_gather.txt 2>&1
cat $O/r5c5_gather.txt
cat $O/r5c5_debug.txt | tail -15
