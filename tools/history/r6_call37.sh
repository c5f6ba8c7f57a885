#!/bin/sh
# round 6, call 37: the up-projection's K loop without its staging instructions / without its fragment reads / without both (timing probes, wrong results)
sh tools/tm_store_ab.sh abstage abfrags abstagefrags
mv gpurun_out/r6_probe_store_policy.txt gpurun_out/r6_probe_t4up_kloop_ablations2.txt
