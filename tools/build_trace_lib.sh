#!/bin/sh
# libpips_trace.so = the product objects with gemm.hip rebuilt under -DPIPS_GEMM_TRACE (tools/gemm_trace.py)
set -e
cd "$(dirname "$0")/../pips_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -DPIPS_GEMM_TRACE -c gemm.hip -o /tmp/gemm_trace.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/libpips_trace.so /tmp/gemm_trace.o encoder.o track.o gather_tiled.o gemm_bf16.o api.o
