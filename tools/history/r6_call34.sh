#!/bin/sh
# round 6, call 34: MFMAs with the LDS reads of their own wave between them, both bf16 shapes at the up-projection's ratio (tools/mfma_lds_mix.hip)
mkdir -p gpurun_out
./tools/mfma_lds_mix > gpurun_out/r6_probe_mfma_lds_mix.txt 2>&1; cat gpurun_out/r6_probe_mfma_lds_mix.txt
