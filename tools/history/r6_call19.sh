#!/bin/sh
# round 6, call 19: what the up-projection's GELU epilogue costs (timing probe: the kernel without the GELU arithmetic, wrong results)
sh tools/tm_store_ab.sh nogelu
mv gpurun_out/r6_probe_store_policy.txt gpurun_out/r6_probe_t4up_without_gelu.txt
