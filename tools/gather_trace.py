"""Stage timeline of gather_tiled_kernel (PIPS_TILED_TRACE build): PIPS_LIB_PATH=build/libpips_trace.so python tools/gather_trace.py"""
import os, sys, ctypes
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
c = grid.reshape(1, N, 1, 2).repeat(B, 1, S, 1).reshape(M, 2).contiguous().to(dev)
nblk = 16384                                 # 256 blocks x 64 items
tr = torch.zeros(nblk, 16, dtype=torch.int64, device=dev)
lib.pips_tiled_trace.argtypes = [ctypes.c_void_p]
for _ in range(2): ops.mixer_input_build_tiled(pyr, B, H8, W8, ffeats, c)
torch.cuda.synchronize()
assert lib.pips_tiled_trace(tr.data_ptr()) == 0
ops.mixer_input_build_tiled(pyr, B, H8, W8, ffeats, c)
torch.cuda.synchronize()
lib.pips_tiled_trace(None)
t = tr.cpu()
t = t[t[:, 2] > 0]
print("items traced:", t.shape[0])
names = ["C++ prologue (item fetch, geometry, DMA(0) issue)", "asm item body (set-up, 8 phases, blend + store)"]
d = (t[:, 1:3] - t[:, 0:2]).float()
tot = (t[:, 2] - t[:, 0]).float()
print("s_memtime ticks per item (mean / p90):  total %.0f / %.0f" % (float(tot.mean()), float(tot.quantile(0.9))))
for i, n in enumerate(names):
    print(f"  {n:50s} {float(d[:, i].mean()):9.0f} {float(d[:, i].quantile(0.9)):9.0f}")
pn = ["", "", "", "", "", "set-up (addresses, masks)", "phases: wait vmcnt(0)", "phases: barrier", "phases: pair 0", "phases: pair 1", "phases: pair 2", "(after phases)", "blend + store"]
for i in range(5, 13):
    print(f"    asm, wave 0: {pn[i]:34s} {float(t[:, i].float().mean()):9.0f} {float(t[:, i].float().quantile(0.9)):9.0f}")
print("  C++ prologue split (wave 0): geometry+dma_setup %.0f | DMA(0) issue %.0f | records/geo %.0f | same+pack %.0f" % (
    float((t[:, 13] - t[:, 0]).float().mean()), float((t[:, 14] - t[:, 13]).float().mean()), float((t[:, 15] - t[:, 14]).float().mean()), float((t[:, 1] - t[:, 15]).float().mean())))
