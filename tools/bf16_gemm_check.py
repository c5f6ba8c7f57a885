"""The config-3 up-projection (bf16 A, bf16 W, GELU, bf16 C) through pips_gemm_bf16 against a torch reference with the same
roundings (fp32 accumulate, bf16 round, exact GELU, bf16 round): where the two differ, by tile / wave / register."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import _tunelib  # noqa: F401  (PIPS_LIB_PATH -> pips_amd._lib.use_library)
from pips_amd import ops
dev = "cuda:0"
M, N, K = int(os.environ.get("M", 16384)), int(os.environ.get("N", 2048)), 512
g = torch.Generator().manual_seed(0)
A = (torch.randn(M, K, generator=g)).to(dev).bfloat16()
W = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev).bfloat16()
b = torch.randn(N, generator=g).to(dev)
out = ops.gemm_bf16(A, W, b, epi=1, out_bf16=True).float()
pre = (A.float() @ W.float().t() + b).bfloat16().float()
ref = torch.nn.functional.gelu(pre).bfloat16().float()
err = (out - ref).abs()
print("max err", float(err.max()), "mean", float(err.mean()), " nan:", int(torch.isnan(out).sum()))
bad = err > 0.05
print("bad fraction", float(bad.float().mean()))
if bad.any():
    bt = bad.reshape(M // 256, 256, N // 128, 128).permute(0, 2, 1, 3)          # (tm, tn, 256, 128)
    per_tile = bt.float().mean(dim=(2, 3))
    print("tiles with errors:", int((per_tile > 0).sum()), "of", per_tile.numel(), " first:", (per_tile > 0).nonzero()[:6].tolist())
    tm, tn = (per_tile > 0).nonzero()[0].tolist()
    t = bt[tm, tn].float()
    print("bad tile", tm, tn, ": by wave (4x2 of 64x64):", t.reshape(4, 64, 2, 64).mean(dim=(1, 3)).tolist())
    w = t[:64, :64]
    print("  wave(0,0) rows with errors:", w.mean(dim=1).nonzero().flatten()[:16].tolist(), " cols:", w.mean(dim=0).nonzero().flatten()[:32].tolist())
    r0 = (tm * 256, tn * 128)
    print("  sample out/ref:", out[r0[0], r0[1]:r0[1] + 8].tolist(), ref[r0[0], r0[1]:r0[1] + 8].tolist())
