#!/bin/sh
# A/B of output-store cache policies in the bf16 mixer pass (tools/tm_store_ab.py): the product library, then the variant libraries named
# on the command line (build/libpips_<name>.so), each under rocprofv3 --kernel-trace --stats and once more without the profiler.
mkdir -p gpurun_out
R="$(pwd)"
cd /tmp && export TMPDIR=/tmp && cd "$R"
OUT=gpurun_out/r6_probe_store_policy.txt
: > $OUT
for v in product "$@" product; do
    if [ $v = product ]; then unset PIPS_LIB_PATH; else export PIPS_LIB_PATH=build/libpips_$v.so; fi
    python tools/tm_store_ab.py 2>/dev/null | grep "ms per mixer pass" >> $OUT
    rm -rf /tmp/tmab && rocprofv3 --kernel-trace --stats -d /tmp/tmab -o r -- python tools/tm_store_ab.py > /tmp/tmab.log 2>&1
    for f in $(find /tmp/tmab -name "*.db"); do python tools/rocpd_summary.py $f /tmp/tmab_stats.txt > /dev/null; done
    grep -E "token_mix|gemm_bf16_t4|gemm_f32_t4|ln_mean" /tmp/tmab_stats.txt | cut -c1-60,110-160 | sed 's/^/    /' >> $OUT
done
cat $OUT
