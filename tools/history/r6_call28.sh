#!/bin/sh
# round 6, call 28: headline step time with and without the fp32 stem change, without the profiler (3 alternations), then per-kernel tables of both
mkdir -p gpurun_out
R="$(pwd)"; cd /tmp && export TMPDIR=/tmp && cd "$R"
O=gpurun_out/r6_probe_stem_f32_v4_steps.txt
: > $O
for k in 1 2 3; do
for v in product prevstem32; do
    L=""; [ $v = product ] || L="--lib build/libpips_$v.so"
    python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-stage-profile --no-extras $L 2>/dev/null > /tmp/su.log
    echo "$v: $(grep -o '"ms_per_step": [0-9.]*' /tmp/su.log | head -1) $(grep -o '"ms_per_step_median": [0-9.]*' /tmp/su.log | head -1)" >> $O
done
done
for v in product prevstem32; do
    L=""; [ $v = product ] || L="--lib build/libpips_$v.so"
    rm -rf /tmp/su && rocprofv3 --kernel-trace --stats -d /tmp/su -o r -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-stage-profile --no-extras $L > /tmp/su.log 2>&1
    for f in $(find /tmp/su -name "*.db"); do python tools/rocpd_summary.py $f gpurun_out/r6c28_stats_$v.txt > /dev/null; done
done
cat $O
