#!/bin/sh
# libpips_trace.so = the library built under -DPIPS_GEMM_TRACE (per-block phase timestamps in the
# fp32 GEMM/conv kernel; read back by tools/gemm_trace.py).  Not a product build.
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
D=/tmp/pips_trace_build; mkdir -p "$D"
cd "$ROOT/pips_amd/csrc"
OBJS=""
for f in *.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -DPIPS_GEMM_TRACE -c "$f" -o "$D/${f%.hip}.o" &
  OBJS="$OBJS $D/${f%.hip}.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/tools/libpips_trace.so" $OBJS
