"""GEMM main-loop ablations (needs tools/libpips_ablate.so built with -DPIPS_GEMM_ABLATE)."""
import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import _tunelib  # noqa: F401  (PIPS_LIB_PATH -> pips_amd._lib.use_library)
from pips_amd import ops
dev = "cuda:0"
def ev(fn, reps):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps
g = torch.Generator().manual_seed(0)
print("tile", os.environ.get("PIPS_GEMM_TILE"))
for (M, N, K) in [(16384, 2048, 512), (16384, 2048, 4096), (2048, 2048, 512), (2048, 512, 2048)]:
    A = torch.randn(M, K, generator=g).to(dev)
    W = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    for name, flags in [("full+gelu", 1), ("full", 0), ("no-load", 0x100), ("no-load no-store", 0x300),
                        ("no-load no-store no-barrier", 0x700), ("no-store", 0x200)]:
        ms = ev(lambda: ops.gemm(A, W, b, flags), 20)
        print(f"M={M} N={N} K={K} {name:30s} {ms*1e3:8.1f} us {2.0*M*N*K/ms/1e9:7.1f} TF", flush=True)
