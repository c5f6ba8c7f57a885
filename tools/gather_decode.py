"""Decode what the tiled gather reads: pyramid value = x + 100*y + 10000*c, one-hot features, integer coords."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pips_amd import ops, _lib
lib = _lib.load()
dev = torch.device("cuda:0")
B, H8, W8, N = 1, 16, 20, 40
c0 = int(sys.argv[1]) if len(sys.argv) > 1 else 5
S, F, M = 8, 8, B * N * 8
pyr = torch.zeros(lib.pips_pyramid_floats(F, H8 * 8, W8 * 8, 8))
for l in range(4):
    h, w = H8 >> l, W8 >> l
    off = lib.pips_pyramid_offset(F, H8 * 8, W8 * 8, 8, l)
    ys, xs, cs = torch.meshgrid(torch.arange(h), torch.arange(w), torch.arange(128), indexing="ij")
    val = (xs + 100 * ys + 10000 * cs).float().reshape(1, -1).repeat(F, 1).reshape(-1)
    pyr[off:off + val.numel()] = val
pyr = pyr.to(dev)
ffeats = torch.zeros(M, 128); ffeats[:, c0] = 128 ** 0.5
g = torch.Generator().manual_seed(0)
xy = torch.stack([torch.randint(4, W8 - 5, (N,), generator=g), torch.randint(4, H8 - 5, (N,), generator=g)], -1).float()
c = xy.reshape(1, N, 1, 2).repeat(B, 1, S, 1).reshape(M, 2).contiguous()
Xt = ops.mixer_input_build_tiled(pyr, B, H8, W8, ffeats.to(dev), c.to(dev), out=torch.full((M, 544), 7.0, device=dev)).cpu()
Xd = ops.mixer_input_build(pyr, B, H8, W8, ffeats.to(dev), c.to(dev)).cpu()
def dec(v):
    v = int(round(float(v)))
    return (v % 100, (v % 10000) // 100, v // 10000)
for m in (0, 8, 17):
    print("row", m, "coord", c[m].tolist())
    for l in range(2):
        t = Xt[m, 128 + l * 49:128 + (l + 1) * 49].view(7, 7)     # [ix][iy]
        d = Xd[m, 128 + l * 49:128 + (l + 1) * 49].view(7, 7)
        print(" level", l)
        for iy in range(7):
            print("   tiled ", [dec(t[ix, iy]) for ix in range(7)])
            print("   direct", [dec(d[ix, iy]) for ix in range(7)])
print("---- per-channel sweep: max |tiled - direct| per level with one-hot features")
for cc in (0, 17, 100):
    ffeats = torch.zeros(M, 128); ffeats[:, cc] = 128 ** 0.5
    Xt = ops.mixer_input_build_tiled(pyr, B, H8, W8, ffeats.to(dev), c.to(dev)).cpu()
    Xd = ops.mixer_input_build(pyr, B, H8, W8, ffeats.to(dev), c.to(dev)).cpu()
    d = (Xt[:, 128:324] - Xd[:, 128:324]).abs().view(M, 4, 49).amax(dim=(0, 2))
    if float(d.max()) > 1e-2 * (1 + 10000 * cc) * 1e-3:
        m = int((Xt[:, 128:324] - Xd[:, 128:324]).abs().view(M, -1).amax(1).argmax())
        l = int(d.argmax())
        k = int((Xt[m, 128 + l * 49:128 + (l + 1) * 49] - Xd[m, 128 + l * 49:128 + (l + 1) * 49]).abs().argmax())
        print(f"channel {cc}: level diffs {[round(float(x), 1) for x in d]}  e.g. row {m} level {l} tap {k}: tiled {dec(Xt[m, 128 + l * 49 + k])} direct {dec(Xd[m, 128 + l * 49 + k])}")
