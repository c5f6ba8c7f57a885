"""Split-bf16 (bf16x3) GEMM / conv: accuracy against fp64 next to the exact-fp32 MFMA path, and timing."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
import _tunelib  # noqa: F401  (PIPS_LIB_PATH -> pips_amd._lib.use_library)
from pips_amd import ops

dev = "cuda:0"
g = torch.Generator().manual_seed(0)
def ev(fn, reps):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps
print("PIPS_X3_TILE =", os.environ.get("PIPS_X3_TILE"))
for (M, N, K, epi) in [(2048, 2048, 512, 1), (2048, 512, 2048, 2), (2048, 512, 544, 0), (256, 1040, 512, 0), (333, 520, 96, 1),
                       (16384, 2048, 512, 1), (16384, 512, 2048, 2)]:
    A = torch.randn(M, K, generator=g).to(dev)
    W = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    R = torch.randn(M, N, generator=g).to(dev) if epi == 2 else None
    W3 = ops.split_bf16x3(W)
    ref = A.double() @ W.double().T + b.double()
    if epi == 1: ref = F.gelu(ref)
    if epi == 2: ref = ref + R.double()
    c32 = ops.gemm(A, W, b, epi, R)
    cx3 = ops.gemm_x3(A, W3, b, epi, R)
    e32 = (c32.double() - ref).abs().max().item(); ex3 = (cx3.double() - ref).abs().max().item()
    t32 = ev(lambda: ops.gemm(A, W, b, epi, R), 30); tx3 = ev(lambda: ops.gemm_x3(A, W3, b, epi, R), 30)
    fl = 2.0 * M * N * K
    print(f"gemm M={M:6d} N={N:5d} K={K:5d} epi={epi}: max|err| fp32-mfma {e32:.2e}  x3 {ex3:.2e} | "
          f"{t32*1e3:7.1f} us ({fl/t32/1e9:6.1f} TF) -> {tx3*1e3:7.1f} us ({fl/tx3/1e9:6.1f} TF-equiv)", flush=True)

for (F_, H, Wd, Cin, Cout, k, s) in [(2, 37, 45, 64, 64, 3, 1), (8, 184, 248, 64, 64, 3, 1), (8, 184, 248, 64, 96, 3, 2), (8, 92, 124, 96, 96, 3, 1),
                                    (8, 92, 124, 96, 128, 3, 2), (8, 46, 62, 128, 128, 3, 1), (8, 23, 31, 128, 128, 3, 1),
                                    (8, 46, 62, 416, 256, 3, 1), (8, 46, 62, 256, 128, 1, 1)]:
    x = torch.randn(F_, H, Wd, Cin, generator=g).to(dev)
    w = (torch.randn(Cout, k, k, Cin, generator=g) / math.sqrt(Cin * k * k)).to(dev)
    b = torch.randn(Cout, generator=g).to(dev)
    p = 1 if k == 3 else 0
    w3 = ops.split_bf16x3(w)
    ref = F.conv2d(x.permute(0, 3, 1, 2).double(), w.permute(0, 3, 1, 2).double(), b.double(), stride=s, padding=p).permute(0, 2, 3, 1)
    o32, s32 = ops.conv_nhwc(x, w, b, k, s, p, want_stats=True)
    ox3, sx3 = ops.conv_nhwc_x3(x, w3, b, k, s, p, want_stats=True)
    e32 = (o32.double() - ref).abs().max().item(); ex3 = (ox3.double() - ref).abs().max().item()
    # statistics partials: totals per (frame, channel) must agree with the output
    tot = ox3.double().sum(dim=(1, 2)); st = sx3.double().sum(dim=1)[..., 0]
    es = ((tot - st).abs().max() / tot.abs().max()).item()
    t32 = ev(lambda: ops.conv_nhwc(x, w, b, k, s, p, want_stats=False), 10); tx3 = ev(lambda: ops.conv_nhwc_x3(x, w3, b, k, s, p, want_stats=False), 10)
    Ho, Wo = ref.shape[1], ref.shape[2]
    fl = 2.0 * F_ * Ho * Wo * Cout * Cin * k * k
    print(f"conv {F_}x{H}x{Wd} {Cin}->{Cout} k{k} s{s}: max|err| fp32-mfma {e32:.2e}  x3 {ex3:.2e} stats-rel {es:.1e} | "
          f"{t32*1e3:7.1f} us ({fl/t32/1e9:6.1f} TF) -> {tx3*1e3:7.1f} us ({fl/tx3/1e9:6.1f} TF-equiv)", flush=True)
