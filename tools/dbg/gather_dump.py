import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pips_amd import ops, _lib
lib = _lib.load()
dev = torch.device("cuda:0")
B, H8, W8, N = 1, 46, 62, 300
S, F, M = 8, B * 8, B * N * 8
g = torch.Generator().manual_seed(0)
pyr = torch.randn(lib.pips_pyramid_floats(F, H8 * 8, W8 * 8, 8), generator=g).to(dev)
ffeats = torch.randn(M, 128, generator=g).to(dev)
c = (torch.rand(M, 2, generator=g) * torch.tensor([W8 - 1.0, H8 - 1.0])).to(dev)
X = torch.zeros(M, 544, device=dev)
tt = ops.times_table(dev)
nb = lib.pips_gather_scratch_bytes(B, N, H8, W8)
scratch = torch.zeros(nb, dtype=torch.uint8, device=dev)
rc = lib.pips_mixer_input_build_tiled(_lib.ptr(pyr), B, S, H8, W8, _lib.ptr(ffeats), _lib.ptr(c), _lib.ptr(tt), N, _lib.ptr(X), _lib.ptr(scratch), nb, ops._stream())
torch.cuda.synchronize()
print("rc", rc, "scratch bytes", nb)
sc = scratch.cpu().view(torch.int32)
rec = sc[:F * N * 16].reshape(F, N, 4, 4)
print("rec[0,0]:", rec[0, 0].tolist())
print("rec float view wx wy:", rec[0, 0, :, 1:3].view(torch.float32).tolist())
tiles = ((W8 + 15) // 16) * ((H8 + 15) // 16); max_items = tiles + N // 96 + 1
off_items = ((F * N * 64 + 255) // 256 * 256) // 4
items = sc[off_items: off_items + F * max_items * 4].reshape(F, max_items, 4)
print("max_items", max_items, "items[0,:4]:", items[0, :4].tolist())
off_igeo = off_items + ((F * max_items * 16 + 255) // 256 * 256) // 4
ig = sc[off_igeo: off_igeo + F * max_items * 16].reshape(F, max_items, 4, 4)
print("igeo[0,0]:", [[v & 0xffff, v >> 16] for v in ig[0, 0, :, 0].tolist()], [[v & 0xffff, v >> 16] for v in ig[0, 0, :, 1].tolist()], ig[0, 0, :, 3].tolist())
print("X corr nan count", int(torch.isnan(X[:, 128:324]).sum()), "of", X[:, 128:324].numel(), "rows with nan", int(torch.isnan(X[:, 128:324]).any(1).sum()))
Xd = ops.mixer_input_build(pyr, B, H8, W8, ffeats, c)
d = (X[:, 128:324] - Xd[:, 128:324]).abs()
print("max diff where finite", float(d[~torch.isnan(d)].max()))
