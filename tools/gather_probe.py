"""Correlation-gather probe at an arbitrary geometry (default: BASELINE config 4, B=4 720x1280 N=4096 grid):
checks the LDS-tiled kernel against the direct one and times both.
usage: python tools/gather_probe.py [B H8 W8 N] [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pips_amd import ops, _lib
lib = _lib.load()
dev = torch.device("cuda:0")
a = [int(x) for x in sys.argv[1:5]] if len(sys.argv) >= 5 else [4, 90, 160, 4096]
B, H8, W8, N = a
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 20
S, F, M = 8, B * 8, B * N * 8
g = torch.Generator().manual_seed(0)
pyr = torch.randn(lib.pips_pyramid_floats(F, H8 * 8, W8 * 8, 8), generator=g).to(dev)
ffeats = torch.randn(M, 128, generator=g).to(dev)
n = int(round(N ** 0.5))
gy, gx = torch.meshgrid(torch.linspace(1, H8 - 2, n), torch.linspace(1, W8 - 2, n), indexing="ij")
grid = torch.stack([gx.reshape(-1), gy.reshape(-1)], -1)[:N]                       # (N,2) map px
base = grid.reshape(1, N, 1, 2).repeat(B, 1, S, 1).reshape(M, 2).contiguous().to(dev)
lv = sum((H8 >> l) * (W8 >> l) for l in range(4))
comp = F * (lv * 512 + N * 512 + N * 8 + N * 196 * 4)
gath = M * (4 * 64 * 512 + 512 + 8 + 196 * 4)

def ev(fn, reps):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps

for jitter in (0.0, 0.5, 2.0, 8.0):
    c = (base + torch.randn(M, 2, generator=g).to(dev) * jitter).contiguous()
    Xt = ops.mixer_input_build_tiled(pyr, B, H8, W8, ffeats, c)
    Xd = ops.mixer_input_build(pyr, B, H8, W8, ffeats, c)
    err = float((Xt[:, 128:324] - Xd[:, 128:324]).abs().max())
    same_rest = bool(torch.equal(Xt[:, :128], Xd[:, :128]) and torch.equal(Xt[:, 324:], Xd[:, 324:]))
    tt = ev(lambda: ops.mixer_input_build_tiled(pyr, B, H8, W8, ffeats, c), reps)
    td = ev(lambda: ops.mixer_input_build(pyr, B, H8, W8, ffeats, c), max(reps // 4, 2))
    print(f"jitter {jitter:4.1f} px: tiled(bin+embed+gather) {tt*1e3:8.1f} us  direct {td*1e3:8.1f} us   "
          f"max|corr diff| {err:.2e} rest-equal {same_rest}   compulsory {comp/1e6:.1f} MB -> "
          f"{comp/tt/1e6:.0f} GB/s tiled-total ({comp/tt/1e6/80:.1f}% of 8 TB/s)", flush=True)
