"""The tiled gather at BASELINE configs[3] geometry (B=4, 90x160 maps, N=4096 on a 64x64 grid): HIP-event time of its three
launches (bin / embed / gather) on synthetic maps -- the quick form of bench.py's config4 gather_roofline (no encoder)."""
import os, sys, statistics
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
ops.pyramid_mirror(pyr, F, H8 * 8, W8 * 8, 8)                 # the bf16 mode's kernel reads the bf16 mirror
lv = sum((H8 >> l) * (W8 >> l) for l in range(4))
for name, c in (("grid", grid.reshape(1, N, 1, 2).repeat(B, 1, S, 1)),
                ("grid + 2 px noise", grid.reshape(1, N, 1, 2).repeat(B, 1, S, 1) + torch.randn(B, N, S, 2, generator=g) * 2)):
    c = c.reshape(M, 2).contiguous().to(dev)
    for bf in (False, True):
        ts = {"bin": [], "embed": [], "gather": []}
        for i in range(14):
            _, t = ops.mixer_input_build_tiled_timed(pyr, B, H8, W8, ffeats, c, bf16_maps=bf)
            if i >= 4:
                for k in ts: ts[k].append(t[k])
        # pyramid + features + coordinates + fcorrs; bf16 mode: bf16 pyramid and features, fp32 out
        comp = F * (lv * (256 if bf else 512) + N * (256 if bf else 512) + N * 8 + N * 196 * 4)
        tg = statistics.mean(ts["gather"])
        print(f"{name}, {'bf16 mode (gather_mfma_kernel)' if bf else 'fp32 (gather_tiled_kernel)'}: bin {statistics.mean(ts['bin'])*1e3:.1f} us  "
              f"embed {statistics.mean(ts['embed'])*1e3:.1f} us  gather {tg*1e3:.1f} us = {comp/1e6:.1f} MB / launch = {comp/tg/1e6:.0f} GB/s "
              f"compulsory = {comp/tg/1e6/8000:.3f} of 8 TB/s")
# BASELINE configs[2] geometry (B=8, 46x62 maps, N=256: ~21 particles per tile): the direct bf16-map kernel against the tiled matrix-core path
B3, H3, W3, N3 = 8, 46, 62, 256
F3, M3 = B3 * 8, B3 * N3 * 8
pyr3 = ops.pyramid_mirror(torch.randn(lib.pips_pyramid_floats(F3, H3 * 8, W3 * 8, 8), generator=g).to(dev), F3, H3 * 8, W3 * 8, 8)
ff3 = torch.randn(M3, 128, generator=g).to(dev)
c3 = (torch.rand(M3, 2, generator=g) * torch.tensor([W3 - 1.0, H3 - 1.0])).to(dev)
ev = lambda fn, n=20: (lambda e0, e1: (e0.record(), [fn() for _ in range(n)], e1.record(), e1.synchronize(), e0.elapsed_time(e1) / n)[-1])(
    torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
ops.mixer_input_build(pyr3, B3, H3, W3, ff3, c3, bf16_maps=True)
print(f"config-3 geometry: direct bf16-map kernel {ev(lambda: ops.mixer_input_build(pyr3, B3, H3, W3, ff3, c3, bf16_maps=True))*1e3:.1f} us", end="")
ts = {"bin": [], "embed": [], "gather": []}
for i in range(14):
    _, t = ops.mixer_input_build_tiled_timed(pyr3, B3, H3, W3, ff3, c3, bf16_maps=True)
    if i >= 4:
        for k in ts: ts[k].append(t[k])
print("; tiled matrix-core path: bin %.1f + embed %.1f + gather %.1f us" % tuple(statistics.mean(ts[k]) * 1e3 for k in ("bin", "embed", "gather")))
