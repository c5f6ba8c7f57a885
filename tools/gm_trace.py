"""Per-step timeline of gather_mfma_kernel (a -DGM_TRACE build, PIPS_LIB_PATH): s_memtime stamps of the product wave 0 and the loader wave 12
of blocks 0 and 1 at config-4 geometry.  Prints the mean clocks between consecutive stamps by (from tag -> to tag).
tags: 1 batch start | loader: 12 next item, 20 step top, 21 element delivered, 22 next element requested, 23 barrier passed |
product: 30 item start, 43 blend done, 44 barrier passed, 40 step top, 45 before the products, 46 window test done, 41 MFMAs done, 42 scatter done.
A stamp costs ~160 clocks; a stamp right behind another one also waits for its store (flat: it counts in lgkmcnt) -- ~700 clocks."""
import os, sys, ctypes, collections
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
TRN = 8192
for _ in range(3):
    ops.mixer_input_build_tiled(pyr, B, H8, W8, ffeats, c, bf16_maps=True)
buf = torch.zeros(4 * TRN, dtype=torch.int64, device=dev)
lib._handle if False else None
fn = ctypes.CDLL(_lib.LIB_PATH).pips_gm_trace
fn.argtypes = [ctypes.c_void_p]
assert fn(ctypes.c_void_p(buf.data_ptr())) == 0
_, t = ops.mixer_input_build_tiled_timed(pyr, B, H8, W8, ffeats, c, bf16_maps=True)
torch.cuda.synchronize()
print("gather (traced build): %.1f us" % (t["gather"] * 1e3))
tr = buf.cpu().numpy().reshape(4, TRN)
for w, name in enumerate(("block 0 product wave 0", "block 0 loader wave 12", "block 1 product wave 0", "block 1 loader wave 12")):
    v = [int(x) for x in tr[w] if x != 0]
    ts = [(x >> 8, x & 0xff) for x in v]
    if len(ts) < 2:
        print(name, "no stamps"); continue
    total = ts[-1][0] - ts[0][0]
    acc = collections.OrderedDict()
    for (t0, a), (t1, b) in zip(ts[:-1], ts[1:]):
        k = (a, b)
        e = acc.setdefault(k, [0, 0])
        e[0] += t1 - t0; e[1] += 1
    items = sum(1 for _, a in ts if a == 1)
    print(f"{name}: {len(ts)} stamps, {items} items, first -> last {total} ticks")
    for (a, b), (s_, n_) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
        print(f"   {a:3d} -> {b:3d}: n {n_:5d}  mean {s_ / n_:9.1f}  total {s_:10d}  ({100.0 * s_ / total:5.1f} %)")
