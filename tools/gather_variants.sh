#!/bin/sh
# Build the tuning variants of the tiled gather (build/libpips_*.so): parts of the generated item body left out
# (wrong results, timing only) and the traced build.  Run on the build box, then on the GPU box
#   for v in build/libpips_no*.so; do PIPS_LIB_PATH=$v python tools/gather_ablate.py; done
#   PIPS_LIB_PATH=build/libpips_trace.so python tools/gather_trace.py
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
cd "$ROOT"
for a in reads feats fma dma warm epi; do
  mkdir -p build/inc_$a
  PIPS_GEN_ABLATE=$a PIPS_GEN_OUT=build/inc_$a/gather_item_asm.inc python tools/gen_gather_asm.py > /dev/null
  sh tools/build_gather_variant.sh no$a "-DPIPS_ITEM_INC=\"$ROOT/build/inc_$a/gather_item_asm.inc\"" > /dev/null
done
sh tools/build_gather_variant.sh noasm -DPIPS_TILED_ABLATE=128 > /dev/null
mkdir -p build/inc_trace
PIPS_GEN_TRACE=1 PIPS_GEN_OUT=build/inc_trace/gather_item_asm.inc python tools/gen_gather_asm.py > /dev/null
sh tools/build_gather_variant.sh trace -DPIPS_TILED_TRACE "-DPIPS_TRACE_WAVE=${PIPS_TRACE_WAVE:-0}" "-DPIPS_ITEM_INC=\"$ROOT/build/inc_trace/gather_item_asm.inc\"" > /dev/null
ls build/*.so
