#!/bin/sh
# round 6, call 21: state_update with the weight column in registers and a particle loop (product) against the one-particle-per-block form
# (build/libpips_prevsu.so) in the configs[2] forward; then the whole GPU suite
mkdir -p gpurun_out
R="$(pwd)"; cd /tmp && export TMPDIR=/tmp && cd "$R"
O=gpurun_out/r6_probe_state_update.txt
: > $O
for v in product prevsu product; do
    L=""; [ $v = product ] || L="--lib build/libpips_$v.so"
    rm -rf /tmp/su && rocprofv3 --kernel-trace --stats -d /tmp/su -o r -- python bench.py --config 3 --steps 5 --warmup 2 --no-cpu-baseline --no-stage-profile --no-extras $L > /tmp/su.log 2>&1
    for f in $(find /tmp/su -name "*.db"); do python tools/rocpd_summary.py $f /tmp/su_stats.txt > /dev/null; done
    echo "$v: $(grep -o '"ms_per_step": [0-9.]*' /tmp/su.log | head -1)" >> $O
    grep -E "state_update|ln_mean|token_mix_mfma" /tmp/su_stats.txt | cut -c1-60,110-160 | sed 's/^/    /' >> $O
done
cat $O
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
