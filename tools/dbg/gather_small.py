import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pips_amd import ops, _lib
lib = _lib.load()
dev = torch.device("cuda:0")
B, H8, W8, N = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
S, F, M = 8, B * 8, B * N * 8
g = torch.Generator().manual_seed(0)
pyr = torch.randn(lib.pips_pyramid_floats(F, H8 * 8, W8 * 8, 8), generator=g).to(dev)
ffeats = torch.randn(M, 128, generator=g).to(dev)
c = (torch.rand(M, 2, generator=g) * torch.tensor([W8 - 1.0, H8 - 1.0])).to(dev)
print("launch", flush=True)
X = ops.mixer_input_build_tiled(pyr, B, H8, W8, ffeats, c)
torch.cuda.synchronize()
print("ok", float(X[:, 128:324].abs().max()), flush=True)
Xd = ops.mixer_input_build(pyr, B, H8, W8, ffeats, c)
print("max diff vs direct", float((X[:, 128:324] - Xd[:, 128:324]).abs().max()))
d = (X[:, 128:324] - Xd[:, 128:324]).abs()
bad = (d > 1e-4).any(1).nonzero().flatten()
print("bad rows", bad.numel(), "of", d.shape[0], "first", bad[:12].tolist())
if bad.numel():
    r = int(bad[0]); k = (d[r] > 1e-4).nonzero().flatten()
    print("row", r, "n,s =", r // 8, r % 8, "bad taps", k.numel(), k[:20].tolist(), "coord", c[r].tolist())
