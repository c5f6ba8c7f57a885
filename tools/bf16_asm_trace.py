"""Timeline of gemm_bf16_gelu_asm_kernel (traced build): s_memtime stamps of one wave of block 0.
   PIPS_GEN_TRACE=1 PIPS_GEN_OUT=build/inc_asmtrace/gemm_bf16_tile_asm.inc python tools/gen_gemm_bf16_asm.py
   sh tools/build_variant.sh asmtrace gemm_bf16_asm -DPIPS_ASM_TRACE '-DPIPS_TILE_INC="<abs path of that .inc>"'
   PIPS_LIB_PATH=build/libpips_asmtrace.so python tools/bf16_asm_trace.py"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import _tunelib  # noqa: F401  (PIPS_LIB_PATH -> pips_amd._lib.use_library)
from pips_amd import ops, _lib
lib = _lib.load()
dev = "cuda:0"
M, N, K = 16384, 2048, 512
g = torch.Generator().manual_seed(0)
A = torch.randn(M, K, generator=g).to(dev).bfloat16()
W = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev).bfloat16()
b = torch.randn(N, generator=g).to(dev)
tr = torch.zeros(8 * 64, dtype=torch.int32, device=dev)
lib.pips_asm_trace.argtypes = [ctypes.c_void_p]
for _ in range(3): ops.gemm_bf16(A, W, b, epi=1, out_bf16=True)
torch.cuda.synchronize()
assert lib.pips_asm_trace(tr.data_ptr()) == 0
ops.gemm_bf16(A, W, b, epi=1, out_bf16=True)
torch.cuda.synchronize()
lib.pips_asm_trace(None)
t = tr.cpu().reshape(8, 64).long() & 0xffffffff
t0 = int(t[0, 0])
for tile in range(4):
    r = t[tile]
    if int(r[0]) == 0: break
    d = lambda a, b: int((r[b] - r[a]) & 0xffffffff)
    print(f"tile {tile}: start +{int((r[0]-t0)&0xffffffff)}  bias wait {d(0,1)}  first reads->ks0 {d(1,2)}")
    for ks in range(8):
        top, midb, w, bar = 2 + 4 * ks, 3 + 4 * ks, 4 + 4 * ks, 5 + 4 * ks
        nxt = 2 + 4 * (ks + 1) if ks < 7 else 34
        print(f"   ks {ks}: a: first step {d(top, 44 + ks):5d} rest {d(44 + ks, midb):5d} | b first half {d(midb, w):5d} | vm wait {d(w, 36 + ks):5d} barrier {d(36 + ks, bar):5d} | b second half {d(bar, nxt):5d}   total {d(top, nxt)}")
    print(f"   park {d(34, 35)}   tile total {d(0, 35)}")
