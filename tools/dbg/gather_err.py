import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pips_amd import ops, _lib
from oracle import pips_oracle as O
sys.path.insert(0, "tests")
lib = _lib.load()
DEV = "cuda:0"
def _grid(n, h, w, margin=8.0):
    k = int(round(n ** 0.5))
    gy, gx = torch.meshgrid(torch.linspace(margin, h - margin, k), torch.linspace(margin, w - margin, k), indexing="ij")
    return torch.stack([gx.reshape(-1), gy.reshape(-1)], -1)
def _pm(t):
    B, S, N, X = t.shape
    return t.permute(0, 2, 1, 3).reshape(B * N * S, X).contiguous()
B, H8, W8, N = 1, 90, 160, 4096
g = torch.Generator().manual_seed(21)
fmaps = torch.randn(B, 8, 128, H8, W8, generator=g)
ffeats = torch.randn(B, 8, N, 128, generator=g)
coords = (_grid(N, H8 * 8, W8 * 8) / 8.0).reshape(1, 1, N, 2).repeat(B, 8, 1, 1) + torch.randn(B, 8, N, 2, generator=g) * 1.5
pyr_ref = O.build_pyramid(fmaps)
ref = torch.cat([O.corr_sample(pyr_ref, ffeats[:, :, n0:n0 + 512], coords[:, :, n0:n0 + 512]) for n0 in range(0, N, 512)], dim=2)
ref64 = torch.cat([O.corr_sample([p.double() for p in pyr_ref], ffeats[:, :, n0:n0 + 512].double(), coords[:, :, n0:n0 + 512].double()) for n0 in range(0, N, 512)], dim=2)
buf = torch.zeros(lib.pips_pyramid_floats(B * 8, H8 * 8, W8 * 8, 8))
for l, p in enumerate(pyr_ref):
    off = lib.pips_pyramid_offset(B * 8, H8 * 8, W8 * 8, 8, l)
    flat = p.reshape(B * 8, 128, p.shape[-2], p.shape[-1]).permute(0, 2, 3, 1).reshape(-1)
    buf[off:off + flat.numel()] = flat
pyr = buf.to(DEV)
ff, co = _pm(ffeats).to(DEV), _pm(coords).to(DEV)
ref_pm = _pm(ref); ref64_pm = _pm(ref64)
Xd = ops.mixer_input_build(pyr, B, H8, W8, ff, co).cpu()[:, 128:324]
Xt = ops.mixer_input_build_tiled(pyr, B, H8, W8, ff, co).cpu()[:, 128:324]
for name, X in (("direct", Xd), ("tiled", Xt), ("oracle32", ref_pm)):
    e = (X.double() - ref64_pm).abs()
    print(name, "vs fp64 oracle: max %.3e" % float(e.max()), "per level max", [float(e[:, l*49:(l+1)*49].max()) for l in range(4)])
e = (Xd - ref_pm).abs()
print("direct vs oracle32 per level", [float(e[:, l*49:(l+1)*49].max()) for l in range(4)])
i = int(e.max(dim=1).values.argmax()); k = int(e[i].argmax())
n, s = i // 8, i % 8
print("worst row", i, "n", n, "s", s, "tap", k, "level", k // 49, "coord", coords[0, s, n].tolist(), "vals", float(Xd[i, k]), float(ref_pm[i, k]), float(ref64_pm[i, k]))
