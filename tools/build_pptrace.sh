#!/bin/sh
# tools/libpips_pptrace.so = the library with -DPIPS_PP_TRACE -DPIPS_TUNING (tools/conv_pp_trace.py).  Not a product build.
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
D=/tmp/pips_pptrace_build; mkdir -p "$D"
cd "$ROOT/pips_amd/csrc"
OBJS=""
for f in *.hip; do
  X=""; [ "$f" = conv_bf16_c64.hip ] && X="-fno-slp-vectorize"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc $X -DPIPS_PP_TRACE -DPIPS_TUNING -c "$f" -o "$D/${f%.hip}.o" 2>/dev/null &
  OBJS="$OBJS $D/${f%.hip}.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/tools/libpips_pptrace.so" $OBJS
