#!/bin/sh
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 900 python -m pytest tests -m gpu -x -q -k "repeatable or batches or mfma" 2>&1 | tail -3 | tee $O/r5c45_tests.txt
