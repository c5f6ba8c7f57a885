#!/bin/sh
# round 6, call 27: the fp32 stem with aligned quad loads, all in flight (product) against build/libpips_prevstem32.so in the headline forward;
# the inorm_apply probe's last word; the GPU suite
mkdir -p gpurun_out
R="$(pwd)"; cd /tmp && export TMPDIR=/tmp && cd "$R"
O=gpurun_out/r6_probe_stem_f32_v4.txt
: > $O
for v in product prevstem32 product prevstem32; do
    L=""; [ $v = product ] || L="--lib build/libpips_$v.so"
    rm -rf /tmp/su && rocprofv3 --kernel-trace --stats -d /tmp/su -o r -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-stage-profile --no-extras $L > /tmp/su.log 2>&1
    for f in $(find /tmp/su -name "*.db"); do python tools/rocpd_summary.py $f /tmp/su_stats.txt > /dev/null; done
    echo "$v: $(grep -o '"ms_per_step": [0-9.]*' /tmp/su.log | head -1)" >> $O
    grep -E "stem_conv" /tmp/su_stats.txt | cut -c1-70,110-160 | sed 's/^/    /' >> $O
done
cat $O
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
