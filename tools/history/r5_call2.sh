#!/bin/sh
# round 5, call 2: the bf16 mode's matrix-core gather -- parity tests, then timing at config-4 / config-3 geometry
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q -s -k "mfma or bf16_matrix or bf16_mode_end_to_end or mixer_input_build_tiled or config4_gather" > $O/r5c2_tests.log 2>&1
echo "tests rc=$?"; grep -v "^$" $O/r5c2_tests.log | tail -25
timeout 300 python tools/gather_c4.py > $O/r5c2_gather.txt 2>&1
cat $O/r5c2_gather.txt
