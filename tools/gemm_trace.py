"""Per-block phase timeline of the fp32 GEMM kernel.

Needs tools/libpips_trace.so (gemm.hip compiled with -DPIPS_GEMM_TRACE, see tools/README in
DESIGN.md §4) and PIPS_LIB_PATH pointing at it.  Every block's thread 0 stamps the 100 MHz
constant clock at entry / after the prologue barrier / after the K loop / after the K-split
reduction / after the epilogue; this script prints how the launch's wall time splits.
usage: PIPS_LIB_PATH=tools/libpips_trace.so python tools/gemm_trace.py M N K epi
       PIPS_LIB_PATH=tools/libpips_trace.so python tools/gemm_trace.py conv F H W Cin Cout k stride
"""
import ctypes as C, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import _tunelib  # noqa: F401  (PIPS_LIB_PATH -> pips_amd._lib.use_library)
from pips_amd import ops, _lib

dev = "cuda:0"
g = torch.Generator().manual_seed(0)
CONV = len(sys.argv) > 1 and sys.argv[1] == "conv"
if CONV:
    F_, H, Wd, Cin, Cout, k, cs = (int(v) for v in sys.argv[2:9])
    x = torch.randn(F_, H, Wd, Cin, generator=g).to(dev)
    w = (torch.randn(Cout, k, k, Cin, generator=g) / math.sqrt(Cin * k * k)).to(dev)
    b = torch.randn(Cout, generator=g).to(dev)
    pad = 1 if k == 3 else 0
    Ho, Wo = (H + 2 * pad - k) // cs + 1, (Wd + 2 * pad - k) // cs + 1
    M, N, K, epi = F_ * Ho * Wo, Cout, Cin * k * k, 0
    def launch():
        ops.conv_nhwc(x, w, b, k, cs, pad, want_stats=True)
else:
    M, N, K, epi = (int(v) for v in sys.argv[1:5]) if len(sys.argv) >= 5 else (2048, 2048, 512, 1)
    A = torch.randn(M, K, generator=g).to(dev)
    W = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    R = torch.randn(M, N, generator=g).to(dev) if epi == 2 else None
    def launch():
        ops.gemm(A, W, b, epi, R)
lib = _lib.load()
lib.pips_trace_read.argtypes = [C.c_void_p, C.c_size_t]
WARM = int(os.environ.get("TRACE_WARM", "5"))      # many back-to-back launches = sustained-load clocks
for _ in range(WARM):
    launch()
torch.cuda.synchronize()
# surround the traced launch with other launches so it sees the in-situ conditions
for _ in range(3):
    launch()
torch.cuda.synchronize()
nblk_guess = 65536
buf = np.zeros((nblk_guess, 8), dtype=np.uint64)
assert lib.pips_trace_read(buf.ctypes.data_as(C.c_void_p), buf.nbytes) == 0
used = buf[:, 0] > 0
t = buf[used].astype(np.int64)
n = len(t)
t0 = t[:, 0].min()
rel = (t[:, :5] - t0) * 10.0 / 1000.0        # us
span = rel[:, 4].max()
print(f"{'conv ' if CONV else ''}M={M} N={N} K={K} epi={epi} tile={os.environ.get('PIPS_GEMM_TILE')}: {n} blocks, "
      f"launch span {span:.2f} us = {2.0 * M * N * K / span / 1e6:.1f} TF")
names = ["entry", "prologue done", "K loop done", "ksplit reduce done", "epilogue done"]
for i, nm in enumerate(names):
    print(f"  {nm:20s} min {rel[:, i].min():7.2f}  median {np.median(rel[:, i]):7.2f}  max {rel[:, i].max():7.2f} us")
d = np.diff(rel, axis=1)
for i, nm in enumerate(["prologue", "K loop", "reduce", "epilogue"]):
    print(f"  phase {nm:10s} median {np.median(d[:, i]):6.2f}  p10 {np.percentile(d[:, i], 10):6.2f}  p90 {np.percentile(d[:, i], 90):6.2f} us")
mhz = t[:, 7] / np.maximum(d[:, 1], 1e-3)          # shader-clock cycles per us of the K loop
print(f"  shader clock during the K loop: median {np.median(mhz):.0f} MHz (p10 {np.percentile(mhz, 10):.0f}, p90 {np.percentile(mhz, 90):.0f})")
xcc = t[:, 6] & 0xf
hw = t[:, 5]
cu = (hw >> 8) & 0xf
se = (hw >> 13) & 0x7
print("  blocks per XCC:", np.bincount(xcc, minlength=8).tolist())
key = xcc * 1000 + se * 16 + cu
u, cnt = np.unique(key, return_counts=True)
print(f"  distinct (xcc,se,cu): {len(u)}, blocks per CU min/max {cnt.min()}/{cnt.max()}")
# how many rounds: blocks whose entry is later than the earliest finish
first_done = rel[:, 4].min()
print(f"  blocks entering after the first block finished: {(rel[:, 0] > first_done).sum()}")
