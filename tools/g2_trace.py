"""Per-phase clocks of gather_mfma2_kernel's steps (a -DG2_TRACE build, PIPS_LIB_PATH): s_memtime differences accumulated in registers over the
launch by the product wave 0 and the aux wave 12 of blocks 0 and 1, written once at the end (nothing is stored inside the pipeline).
product phases: 0 loop bookkeeping since the barrier | 1 table entry, window test | 2 fragment reads + MFMAs | 3 window scatter | 4 (skipped steps: to
the barrier) | 5 barrier wait.   aux phases: 0 bookkeeping | 1 tap stores of the previous blend | 2 requests (records, next-but-one chunk) | 3 blend |
4 wait for the next chunk | 5 barrier wait | 6 the batch's prologue."""
import os, sys, ctypes
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
for _ in range(3):
    ops.mixer_input_build_tiled(pyr, B, H8, W8, ffeats, c, bf16_maps=True)
buf = torch.zeros(4 * 16, dtype=torch.int64, device=dev)
fn = ctypes.CDLL(_lib.LIB_PATH).pips_g2_trace
fn.argtypes = [ctypes.c_void_p]
assert fn(ctypes.c_void_p(buf.data_ptr())) == 0
_, t = ops.mixer_input_build_tiled_timed(pyr, B, H8, W8, ffeats, c, bf16_maps=True)
torch.cuda.synchronize()
print("gather (traced build): %.1f us" % (t["gather"] * 1e3))
tr = buf.cpu().numpy().reshape(4, 16)
for w, name in enumerate(("block 0 product wave 0", "block 0 aux wave 12", "block 1 product wave 0", "block 1 aux wave 12")):
    steps = int(tr[w][8])
    tot = int(sum(tr[w][:8]))
    print(f"{name}: {steps} steps, {tot} clocks in all = {tot / max(steps, 1):.0f} per step;  per step by phase: " +
          "  ".join(f"{i}: {int(tr[w][i]) / max(steps, 1):.0f}" for i in range(8)))
