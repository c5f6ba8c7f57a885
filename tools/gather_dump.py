"""Tools only: the bf16-mode tiled gather at config-4 geometry on seeded inputs -> the correlation block of X saved to argv[1]
(PIPS_LIB_PATH selects the library); `python tools/gather_dump.py --compare a.pt b.pt` prints the largest difference.  Used to check a
variant build of gather_tiled.hip against the product library (the pytest suite always loads the product library)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
if sys.argv[1] == "--compare":
    a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
    d = (a - b).abs()
    print(f"max |a - b| = {d.max().item():.3e} over {a.numel()} taps ({int((d > 0).sum())} differ); max |a| = {a.abs().max().item():.3e}; "
          f"finite: {bool(torch.isfinite(a).all())} / {bool(torch.isfinite(b).all())}")
    bad = (d > 1e-6).reshape(-1, 8, 4, 49).any(-1)          # (row = particle-major (b, n, s); taps = level x 49)
    print("rows x levels that differ:", int(bad.sum()), "of", bad.numel(), "; by level:", bad.sum((0, 1)).tolist(), "; by frame s:", bad.sum((0, 2)).tolist())
    idx = bad.any(-1).any(-1).nonzero().flatten()
    print("first particles (row // 8):", idx[:20].tolist())
    for r in idx[:3].tolist():
        for s_ in range(2):
            for l in range(4):
                m = (d.reshape(-1, 8, 4, 7, 7)[r, s_, l] > 1e-6).int()       # taps k = ix * 7 + iy
                print("particle", r, "frame", s_, "level", l, "differing taps [ix][iy]:", m.tolist())
    sys.exit(0)
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
gy, gx = torch.meshgrid(torch.linspace(-3, H8 + 2, n), torch.linspace(-3, W8 + 2, n), indexing="ij")      # (windows beyond the map edges too)
grid = torch.stack([gx.reshape(-1), gy.reshape(-1)], -1)
c = (grid.reshape(1, N, 1, 2).repeat(B, 1, S, 1) + torch.randn(B, N, S, 2, generator=g) * 2).reshape(M, 2).contiguous().to(dev)
X = ops.mixer_input_build_tiled(pyr, B, H8, W8, ffeats, c, bf16_maps=True)
torch.cuda.synchronize()
torch.save(X[:, 128:128 + 196].float().cpu(), sys.argv[1])
print("saved", sys.argv[1], _lib.LIB_PATH)
