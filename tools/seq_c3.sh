#!/bin/sh
# per-layer launch sequence of the encoder: config-3 leg and headline (rocprofv3 kernel trace -> tools/rocpd_sequence.py)
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd /tmp && rm -rf /tmp/sq_*
rocprofv3 --kernel-trace --stats -d /tmp/sq_c3 -o p -- python $R/bench.py --leg config3 > /dev/null 2>&1
for f in $(find /tmp/sq_c3 -name "*.db"); do python $R/tools/rocpd_sequence.py $f $R/gpurun_out/seq_config3_encoder.txt > /dev/null; done
rocprofv3 --kernel-trace --stats -d /tmp/sq_h -o p -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-stage-profile --no-extras > /dev/null 2>&1
for f in $(find /tmp/sq_h -name "*.db"); do python $R/tools/rocpd_sequence.py $f $R/gpurun_out/seq_headline_encoder.txt > /dev/null; done
cat $R/gpurun_out/seq_config3_encoder.txt
