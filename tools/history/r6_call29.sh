#!/bin/sh
# round 6, call 29: the fp32 inorm_apply with the statistics in registers (product) against the grid-stride form (build/libpips_prevapply32.so), headline forward
mkdir -p gpurun_out
R="$(pwd)"; cd /tmp && export TMPDIR=/tmp && cd "$R"
O=gpurun_out/r6_probe_inorm_apply_f32.txt
: > $O
for v in product prevapply32; do
    L=""; [ $v = product ] || L="--lib build/libpips_$v.so"
    rm -rf /tmp/su && rocprofv3 --kernel-trace --stats -d /tmp/su -o r -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-stage-profile --no-extras $L > /tmp/su.log 2>&1
    for f in $(find /tmp/su -name "*.db"); do python tools/rocpd_summary.py $f /tmp/su_stats.txt > /dev/null; done
    echo "$v (under rocprofv3):" >> $O
    grep -E "inorm_apply_kernel" /tmp/su_stats.txt | cut -c1-56,110-160 | sed 's/^/    /' >> $O
done
for k in 1 2 3; do
for v in product prevapply32; do
    L=""; [ $v = product ] || L="--lib build/libpips_$v.so"
    python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-stage-profile --no-extras $L 2>/dev/null > /tmp/su.log
    echo "$v: $(grep -o '"ms_per_step": [0-9.]*' /tmp/su.log | head -1) $(grep -o '"ms_per_step_median": [0-9.]*' /tmp/su.log | head -1)" >> $O
done
done
cat $O
python -m pytest tests -m gpu -x -q -k "encoder or forward or golden or inorm or conv" 2>&1 | tail -3
