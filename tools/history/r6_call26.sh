#!/bin/sh
# round 6, call 26: inorm_apply_bf16 with the statistics in registers (product) against the grid-stride form (build/libpips_prevapply.so); encoder / bf16 tests
mkdir -p gpurun_out
R="$(pwd)"; cd /tmp && export TMPDIR=/tmp && cd "$R"
O=gpurun_out/r6_probe_inorm_apply2.txt
: > $O
for v in product prevapply product; do
    L=""; [ $v = product ] || L="--lib build/libpips_$v.so"
    rm -rf /tmp/su && rocprofv3 --kernel-trace --stats -d /tmp/su -o r -- python bench.py --config 3 --steps 5 --warmup 2 --no-cpu-baseline --no-stage-profile --no-extras $L > /tmp/su.log 2>&1
    for f in $(find /tmp/su -name "*.db"); do python tools/rocpd_summary.py $f /tmp/su_stats.txt > /dev/null; done
    echo "$v: $(grep -o '"ms_per_step": [0-9.]*' /tmp/su.log | head -1)" >> $O
    grep -E "inorm_apply_bf16" /tmp/su_stats.txt | cut -c1-56,110-160 | sed 's/^/    /' >> $O
done
cat $O
python -m pytest tests -m gpu -x -q -k "encoder or bf16 or config3 or inorm" 2>&1 | tail -3
