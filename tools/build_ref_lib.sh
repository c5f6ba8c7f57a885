#!/bin/sh
# A/B helper: build libpips_hip.so of a git revision into tools/libpips_<name>.so
# usage: tools/build_ref_lib.sh <git-ref> <name>     then PIPS_LIB_PATH=tools/libpips_<name>.so ...
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
REF="$1"; NAME="$2"
D=/tmp/pips_ref_$NAME
rm -rf "$D"; mkdir -p "$D"
git -C "$ROOT" archive "$REF" pips_amd/csrc include | tar -x -C "$D"
cd "$D/pips_amd/csrc"
OBJS=""
for f in *.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -c "$f" -o "${f%.hip}.o" &
  OBJS="$OBJS ${f%.hip}.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/tools/libpips_$NAME.so" $OBJS
ls -la "$ROOT/tools/libpips_$NAME.so"
