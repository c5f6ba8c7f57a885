#!/bin/sh
# round 6, call 35: the up-projection's K loop without its waits on staged loads / without its barriers / without its fragment waits / without all three
# (timing probes of the generator, wrong results): which dependence holds the loop at ~55 % of the MFMA rate?
sh tools/tm_store_ab.sh abvmwait abbarrier abfragwait abvmwaitbarrie
mv gpurun_out/r6_probe_store_policy.txt gpurun_out/r6_probe_t4up_kloop_ablations.txt
