import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import _tunelib  # noqa: F401  (PIPS_LIB_PATH -> pips_amd._lib.use_library)
from pips_amd import ops, _lib
lib = _lib.load()
dev = "cuda:0"
B, H8, W8, N = 1, 46, 62, 256
F = B * 8
pyr = torch.randn(lib.pips_pyramid_floats(F, H8 * 8, W8 * 8, 8), device=dev)
M = B * N * 8
g = torch.Generator().manual_seed(0)
ff = torch.randn(M, 128, generator=g).to(dev)
co = (torch.rand(M, 2, generator=g) * torch.tensor([W8 - 1.0, H8 - 1.0])).to(dev)
for _ in range(3): ops.mixer_input_build(pyr, B, H8, W8, ff, co)
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): ops.mixer_input_build(pyr, B, H8, W8, ff, co)
e1.record(); e1.synchronize()
print(os.environ.get("PIPS_LIB_PATH", "product")[-20:], "gather %.1f us" % (e0.elapsed_time(e1) / 50 * 1e3))
