"""Time of the bf16 mode's tiled gather at config-4 geometry (PIPS_LIB_PATH selects an ablated build: wrong results, timing only)."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import _tunelib  # noqa: F401
from pips_amd import ops, _lib
lib = _lib.load()
dev = torch.device("cuda:0")
B, H8, W8, N = 4, 90, 160, 4096
S, F, M = 8, B * 8, B * N * 8
g = torch.Generator().manual_seed(0)
pyr = ops.pyramid_mirror(torch.randn(lib.pips_pyramid_floats(F, H8 * 8, W8 * 8, 8), generator=g).to(dev), F, H8 * 8, W8 * 8, 8)
ffeats = torch.randn(M, 128, generator=g).to(dev)
n = 64
gy, gx = torch.meshgrid(torch.linspace(1, H8 - 2, n), torch.linspace(1, W8 - 2, n), indexing="ij")
grid = torch.stack([gx.reshape(-1), gy.reshape(-1)], -1)
c = (grid.reshape(1, N, 1, 2).repeat(B, 1, S, 1) + torch.randn(B, N, S, 2, generator=g) * 2).reshape(M, 2).contiguous().to(dev)
ts = []
for i in range(12):
    _, t = ops.mixer_input_build_tiled_timed(pyr, B, H8, W8, ffeats, c, bf16_maps=True)
    if i >= 4:
        ts.append(t["gather"])
print("%s: gather_mfma_kernel %.1f us" % (os.environ.get("PIPS_LIB_PATH", "product").split("/")[-1], statistics.mean(ts) * 1e3))
