"""Stage timeline of gather_tiled_kernel (PIPS_TILED_TRACE build): PIPS_LIB_PATH=build/libpips_trace.so python tools/gather_trace.py"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pips_amd import ops, _lib
lib = _lib.load()
dev = torch.device("cuda:0")
B, H8, W8, N = 4, 90, 160, 4096
S, F, M = 8, B * 8, B * N * 8
g = torch.Generator().manual_seed(0)
pyr = torch.randn(lib.pips_pyramid_floats(F, H8 * 8, W8 * 8, 8), generator=g).to(dev)
ffeats = torch.randn(M, 128, generator=g).to(dev)
n = 64
gy, gx = torch.meshgrid(torch.linspace(1, H8 - 2, n), torch.linspace(1, W8 - 2, n), indexing="ij")
grid = torch.stack([gx.reshape(-1), gy.reshape(-1)], -1)
c = grid.reshape(1, N, 1, 2).repeat(B, 1, S, 1).reshape(M, 2).contiguous().to(dev)
nblk = 8192
tr = torch.zeros(nblk, 16, dtype=torch.int64, device=dev)
lib.pips_tiled_trace.argtypes = [ctypes.c_void_p]
for _ in range(2): ops.mixer_input_build_tiled(pyr, B, H8, W8, ffeats, c)
torch.cuda.synchronize()
assert lib.pips_tiled_trace(tr.data_ptr()) == 0
ops.mixer_input_build_tiled(pyr, B, H8, W8, ffeats, c)
torch.cuda.synchronize()
lib.pips_tiled_trace(None)
t = tr.cpu()
live = t[:, 14] > 0
t = t[live]
print("blocks traced:", int(live.sum()), " particles/item: min %d mean %.1f max %d" % (int(t[:, 15].min()), float(t[:, 15].float().mean()), int(t[:, 15].max())))
names = ["item read+DMA0", "sort", "geometry", "L0 setup", "L0 phases(8)", "L0 finish", "L1 setup", "L1 phases(8)", "L1 finish",
         "L2 setup", "L2 phases(8)", "L2 finish", "L3 setup", "L3 phases(8)", "L3 finish"]
d = (t[:, 1:15] - t[:, 0:14]).float()
tot = (t[:, 14] - t[:, 0]).float()
print("s_memtime ticks per block (mean / p90):  total %.0f / %.0f" % (float(tot.mean()), float(tot.quantile(0.9))))
for i in [3, 6, 9, 12]:
    print(f"  {names[i + 1] if i else 'sort':16s} {float(d[:, i].mean()):9.0f} {float(d[:, i].quantile(0.9)):9.0f}")
span = float(t[:, 14].max() - t[:, 0].min())
print("kernel span (ticks):", span, " sum of block totals / span =", float(tot.sum()) / span, "(concurrent blocks)")
