#!/bin/sh
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q -s -k "mfma or bf16_matrix or bf16_mode_end_to_end" > $O/r5c6_tests.log 2>&1
echo "tests rc=$?"; grep -v "^$" $O/r5c6_tests.log | tail -12
timeout 300 python -u tools/gather_c4.py > $O/r5c6_gather.txt 2>&1
cat $O/r5c6_gather.txt
