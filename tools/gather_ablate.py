"""Tiled-gather total time (bin + embed + gather) of the library in PIPS_LIB_PATH at config-4 geometry."""
import os, sys
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
base = grid.reshape(1, N, 1, 2).repeat(B, 1, S, 1).reshape(M, 2).contiguous().to(dev)
out = []
for jitter in (0.0, 2.0):
    c = (base + torch.randn(M, 2, generator=g).to(dev) * jitter).contiguous()
    fn = lambda: ops.mixer_input_build_tiled(pyr, B, H8, W8, ffeats, c)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); e1.synchronize()
    out.append(e0.elapsed_time(e1) / 20 * 1e3)
print(f"{os.environ.get('PIPS_LIB_PATH', 'product'):32s} jitter0 {out[0]:7.1f} us   jitter2 {out[1]:7.1f} us", flush=True)
