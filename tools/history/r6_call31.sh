#!/bin/sh
# round 6, call 31: the encoder tests with a width that is no multiple of 4 (the stems' fallback staging) and uint8 frames through the bf16 stem
python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "encoder" 2>&1 | tail -8
