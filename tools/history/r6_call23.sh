#!/bin/sh
# round 6, call 23: configs[2]'s 8 clips as two / four concurrent forwards on separate streams against one forward
mkdir -p gpurun_out
python tools/two_stream_probe.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r6_probe_two_streams.txt; cat gpurun_out/r6_probe_two_streams.txt
