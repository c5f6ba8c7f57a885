#!/bin/sh
# round 6, call 20: ln_mean on the bf16 stream, one wave per particle (product) against the block form (build/libpips_prevln.so); then the mixer tests
sh tools/tm_store_ab.sh prevln
mv gpurun_out/r6_probe_store_policy.txt gpurun_out/r6_probe_ln_mean_wave.txt
python -m pytest tests -m gpu -x -q -k "mixer or config3 or bf16" 2>&1 | tail -3
