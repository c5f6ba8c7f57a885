#!/bin/sh
# round 6, call 40: window lengths up to 32 (PIPS_S_MAX 16 -> 32): the golden cases incl. S = 24, the boundary tests
python -m pytest tests/test_forward_gpu.py -m gpu -x -q -k "window or errors_and_signature" -s 2>&1 | grep -E "w24|passed|failed|Error" | tail -8
