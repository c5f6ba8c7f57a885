#!/bin/sh
# build/libpips_<name>.so = the product library with gather_tiled.hip rebuilt under extra -D flags (tuning / debugging).
# usage: tools/build_gather_variant.sh name -DFLAG ...   ; select with PIPS_LIB_PATH=build/libpips_<name>.so
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
NAME="$1"; shift
mkdir -p "$ROOT/build"
cd "$ROOT/pips_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc "$@" -c gather_tiled.hip -o "$ROOT/build/gather_$NAME.o"
OBJS=""
for f in $(python -c "import sys; sys.path.insert(0, '$ROOT'); from pips_amd import _build; print(' '.join(s[:-4] for s in _build.SOURCES if s != 'gather_tiled.hip'))"); do OBJS="$OBJS $f.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/build/libpips_$NAME.so" $OBJS "$ROOT/build/gather_$NAME.o"
echo "$ROOT/build/libpips_$NAME.so"
