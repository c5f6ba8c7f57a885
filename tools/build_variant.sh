#!/bin/sh
# build/libpips_<name>.so = the product library with ONE translation unit rebuilt under extra -D flags (tuning / debugging).
# usage: tools/build_variant.sh name unit -DFLAG ...   (unit: gemm | encoder | track | gather_tiled | ...)
#        select with PIPS_LIB_PATH=build/libpips_<name>.so
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
NAME="$1"; UNIT="$2"; shift; shift
mkdir -p "$ROOT/build"
cd "$ROOT/pips_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc "$@" -c $UNIT.hip -o "$ROOT/build/${UNIT}_$NAME.o"
# with -DPIPS_TUNING among the flags the other units come from the tuning build (python -m pips_amd._build --tuning): the tuning
# hooks (PIPS_TUNE -> tune_env) live in its api unit
EXT=o
case " $* " in *" -DPIPS_TUNING "*) EXT=tune.o ;; esac
OBJS=""
for f in gemm gemm_f32_t4 conv_f32_t4 encoder encoder_bf16 track gather_tiled scoremap gemm_bf16 gemm_bf16_t4 conv_bf16_c64 conv_bf16_t4c gemm_x3 api; do [ $f = $UNIT ] || OBJS="$OBJS $f.$EXT"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/build/libpips_$NAME.so" $OBJS "$ROOT/build/${UNIT}_$NAME.o"
echo "$ROOT/build/libpips_$NAME.so"
