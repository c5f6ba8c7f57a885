"""Build libpips_hip.so in-tree with hipcc for gfx950 (no JIT cache: the .so must travel
with the source snapshot to the GPU box)."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libpips_hip.so")
SOURCES = ["gemm.hip", "gemm_f32_t4.hip", "conv_f32_t4.hip", "encoder.hip", "encoder_bf16.hip", "track.hip", "gather_tiled.hip", "scoremap.hip", "gemm_bf16.hip", "gemm_bf16_t4.hip", "conv_bf16_c64.hip", "conv_bf16_t4c.hip", "gemm_x3.hip", "api.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function", "-Werror=inline-asm"]
# per-file additions.  conv_bf16_c64.hip: its VALU work runs beside MFMAs of a co-resident wave, where packed fp32 ops are an
# anti-lever -- keep hipcc's SLP vectoriser from re-packing the scalar ops (MI355X_MICROARCH.md)
# track.hip: no IEEE-mode NaN quieting -- under the default (amdgpu-ieee) every fminf / fmaxf gets a `v_max x, x` in front of it to quiet a
# signalling NaN: 512 of the 3 600 vector instructions of a token_mix_mfma_kernel wave, which is bound by exactly those (the GELU of its
# 16 384 hidden units per particle; tools/token_trace_bf16.py).  Results on non-NaN data are bit-identical (the whole GPU suite runs on it)
FILE_FLAGS = {"conv_bf16_c64.hip": ["-fno-slp-vectorize"], "track.hip": ["-fno-honor-nans", "-mno-amdgpu-ieee"]}


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found; libpips_hip.so cannot be built")
    return exe


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def headers():
    """Files every translation unit depends on (beyond its own source)."""
    return [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "gemm_tail.h"), os.path.join(CSRC, "gather_item_asm.inc"), os.path.join(CSRC, "gemm_bf16_t4_asm.inc"), os.path.join(CSRC, "gemm_bf16_t4up_asm.inc"), os.path.join(CSRC, "conv_bf16_t4c_asm.inc"), os.path.join(CSRC, "gemm_f32_t4_asm.inc"), os.path.join(CSRC, "conv_f32_t4_asm.inc"),
            os.path.join(HERE, "..", "include", "pips_hip.h")]


def build_library(force: bool = False, verbose: bool = True, tuning: bool = False) -> str:
    """tuning=True: libpips_hip_tune.so with -DPIPS_TUNING (the PIPS_* environment hooks of common.h are live); objects
    go to csrc/*.tune.o.  The product library has no environment hooks."""
    hipcc = _hipcc()
    headers_ = headers()
    missing = [h for h in headers_ if not os.path.exists(h)]
    if missing:
        raise RuntimeError("pips_amd._build: dependency list names files that do not exist: " + ", ".join(missing))
    objs = []
    procs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src.replace(".hip", ".tune.o" if tuning else ".o"))
        objs.append(o)
        if force or _stale(o, [s] + headers_):
            cmd = [hipcc, *FLAGS, *FILE_FLAGS.get(src, []), *(["-DPIPS_TUNING"] if tuning else []), "-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd))
    lib = LIB.replace(".so", "_tune.so") if tuning else LIB
    if force or _stale(lib, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return lib


if __name__ == "__main__":
    build_library(force="--force" in sys.argv, tuning="--tuning" in sys.argv)
