"""The tiled gather at BASELINE configs[3] geometry (B=4, 90x160 maps, N=4096 on a 64x64 grid): HIP-event time of its three
launches (bin / embed / gather) on synthetic maps -- the quick form of bench.py's config4 gather_roofline (no encoder)."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import _tunelib  # noqa: F401  (PIPS_LIB_PATH -> pips_amd._lib.use_library)
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
for name, c in (("grid", grid.reshape(1, N, 1, 2).repeat(B, 1, S, 1)),
                ("grid + 2 px noise", grid.reshape(1, N, 1, 2).repeat(B, 1, S, 1) + torch.randn(B, N, S, 2, generator=g) * 2)):
    c = c.reshape(M, 2).contiguous().to(dev)
    ts = {"bin": [], "embed": [], "gather": []}
    for i in range(14):
        _, t = ops.mixer_input_build_tiled_timed(pyr, B, H8, W8, ffeats, c)
        if i >= 4:
            for k in ts: ts[k].append(t[k])
    lv = sum((H8 >> l) * (W8 >> l) for l in range(4))
    comp = F * (lv * 512 + N * 512 + N * 8 + N * 196 * 4)
    tg = statistics.mean(ts["gather"])
    print(f"{name}: bin {statistics.mean(ts['bin'])*1e3:.1f} us  embed {statistics.mean(ts['embed'])*1e3:.1f} us  gather {tg*1e3:.1f} us"
          f" = {comp/tg/1e6:.0f} GB/s compulsory = {comp/tg/1e6/8000:.3f} of 8 TB/s")
