"""Micro-benchmark of the fp32-MFMA GEMM / conv kernels (HIP events, rotating weights)."""
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
GEMM_ONLY = "--gemm-only" in sys.argv
print("PIPS_GEMM_TILE =", os.environ.get("PIPS_GEMM_TILE"))
SHAPES = [(2048, 2048, 512, 1), (2048, 512, 2048, 2), (16384, 2048, 512, 1), (16384, 512, 2048, 2)] if GEMM_ONLY else [(2048, 2048, 512, 1), (2048, 512, 2048, 2), (2048, 512, 544, 0), (256, 1040, 512, 0),
                       (16384, 2048, 512, 1), (16384, 512, 2048, 2), (131072, 2048, 512, 1)]
for (M, N, K, epi) in SHAPES:
    A = torch.randn(M, K, generator=g).to(dev)
    Ws = [(torch.randn(N, K, generator=g) / math.sqrt(K)).to(dev) for _ in range(6)]
    b = torch.randn(N, generator=g).to(dev)
    R = torch.randn(M, N, generator=g).to(dev) if epi == 2 else None
    i = [0]
    def run():
        ops.gemm(A, Ws[i[0] % 6], b, epi, R); i[0] += 1
    ms = ev(run, 30)
    print(f"gemm M={M:6d} N={N:5d} K={K:5d} epi={epi}: {ms*1e3:8.1f} us  {2.0*M*N*K/ms/1e9:7.1f} TF", flush=True)

CONVS = [] if GEMM_ONLY else [(8, 184, 248, 64, 64, 3, 1), (8, 184, 248, 64, 96, 3, 2), (8, 92, 124, 96, 96, 3, 1),
                                    (8, 92, 124, 96, 128, 3, 2), (8, 46, 62, 128, 128, 3, 1), (8, 23, 31, 128, 128, 3, 1),
                                    (8, 46, 62, 416, 256, 3, 1), (8, 46, 62, 256, 128, 1, 1)]
for (F_, H, W, Cin, Cout, k, s) in CONVS:
    x = torch.randn(F_, H, W, Cin, generator=g).to(dev)
    w = (torch.randn(Cout, k, k, Cin, generator=g) / math.sqrt(Cin * k * k)).to(dev)
    b = torch.randn(Cout, generator=g).to(dev)
    p = 1 if k == 3 else 0
    ms = ev(lambda: ops.conv_nhwc(x, w, b, k, s, p, want_stats=True), 20)
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    fl = 2.0 * F_ * Ho * Wo * Cout * Cin * k * k
    print(f"conv {H}x{W} {Cin}->{Cout} k{k} s{s}: {ms*1e3:8.1f} us  {fl/ms/1e9:7.1f} TF", flush=True)
