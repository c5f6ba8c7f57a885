"""Where does the tiled gather differ from the direct one?  usage: python tools/gather_debug.py [B H8 W8 N jitter]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pips_amd import ops, _lib
lib = _lib.load()
dev = torch.device("cuda:0")
a = sys.argv[1:]
B, H8, W8, N = [int(x) for x in a[:4]] if len(a) >= 4 else (1, 46, 62, 300)
jit = float(a[4]) if len(a) > 4 else 0.7
S, F, M = 8, B * 8, B * N * 8
g = torch.Generator().manual_seed(0)
pyr = torch.randn(lib.pips_pyramid_floats(F, H8 * 8, W8 * 8, 8), generator=g).to(dev)
ffeats = torch.randn(M, 128, generator=g).to(dev)
n = int(N ** 0.5)
gy, gx = torch.meshgrid(torch.linspace(1, H8 - 2, n), torch.linspace(1, W8 - 2, n), indexing="ij")
grid = torch.stack([gx.reshape(-1), gy.reshape(-1)], -1)
grid = torch.cat([grid, torch.rand(N - grid.shape[0], 2, generator=g) * torch.tensor([W8 - 1.0, H8 - 1.0])])[:N]
c = (grid.reshape(1, N, 1, 2).repeat(B, 1, S, 1).reshape(M, 2) + torch.randn(M, 2, generator=g) * jit).contiguous().to(dev)
Xt = ops.mixer_input_build_tiled(pyr, B, H8, W8, ffeats, c, out=torch.full((M, 544), 7.0, device=dev)).cpu()
print('untouched corr entries:', int((Xt[:, 128:324] == 7.0).sum()), 'of', M * 196, ' zeros:', int((Xt[:, 128:324] == 0).sum()))
Xd = ops.mixer_input_build(pyr, B, H8, W8, ffeats, c).cpu()
d = (Xt[:, 128:324] - Xd[:, 128:324]).abs().view(M, 4, 49)
print(os.environ.get("PIPS_LIB_PATH", "product"), f"B={B} {H8}x{W8} N={N} jitter={jit}")
for l in range(4):
    bad = d[:, l] > 1e-4
    print(f"  level {l}: max diff {float(d[:, l].max()):.3e}  wrong entries {int(bad.sum())}/{bad.numel()}  wrong rows {int(bad.any(1).sum())}/{M}")
rows = (d.view(M, -1) > 1e-4).any(1).nonzero().flatten()
if len(rows):
    m = int(rows[0]); print("  first wrong row", m, "coord", c[m].cpu().tolist())
    for l in range(4):
        print("   tiled ", [round(float(v), 3) for v in Xt[m, 128 + l * 49:128 + l * 49 + 8]])
        print("   direct", [round(float(v), 3) for v in Xd[m, 128 + l * 49:128 + l * 49 + 8]])
