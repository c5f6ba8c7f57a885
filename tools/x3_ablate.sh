#!/bin/sh
# Timing-only ablations of the split-bf16 main loop: builds tools/libpips_x3abl<mask>.so for
# mask in "$@" (1 no split VALU, 2 no global loads, 4 no ds_write, 8 no barrier, 16 every load re-reads K block 0; results invalid).
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
cd "$ROOT/pips_amd/csrc"
for m in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -DPIPS_X3_ABL=$m -c gemm_x3.hip -o /tmp/gemm_x3_abl$m.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/tools/libpips_x3abl$m.so" gemm.o encoder.o track.o gather_tiled.o gemm_bf16.o /tmp/gemm_x3_abl$m.o api.o
done
