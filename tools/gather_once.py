"""A few launches of the tiled gather at config-4 geometry (for rocprofv3 counter passes)."""
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
c = (grid.reshape(1, N, 1, 2).repeat(B, 1, S, 1).reshape(M, 2) + torch.randn(M, 2, generator=g) * float(sys.argv[1] if len(sys.argv) > 1 else 0)).contiguous().to(dev)
bf = len(sys.argv) > 2 and sys.argv[2] == "bf16"          # the bf16 mode's matrix-core kernel on the bf16 mirror
if bf:
    ops.pyramid_mirror(pyr, F, H8 * 8, W8 * 8, 8)
for _ in range(3):
    ops.mixer_input_build_tiled(pyr, B, H8, W8, ffeats, c, bf16_maps=bf)
torch.cuda.synchronize()
