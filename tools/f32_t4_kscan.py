"""Fixed cost and K-loop rate of the fp32 channel-mix GEMM kernels: time (M, N, K) over a range of K, fit t = t0 + K / rate.
usage: python tools/f32_t4_kscan.py [M]     (PIPS_LIB_PATH + PIPS_F32_T4=0/1 select the kernels)"""
import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import _tunelib  # noqa: F401
from pips_amd import ops
dev = "cuda:0"
M = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
g = torch.Generator().manual_seed(0)


def ev(fn, reps):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for (N, epi, Ks) in ((2048, 1, (128, 256, 512, 1024, 2048)), (512, 2, (256, 512, 1024, 2048, 4096))):
    pts = []
    for K in Ks:
        A = torch.randn(M, K, generator=g).to(dev)
        Ws = [(torch.randn(N, K, generator=g) / math.sqrt(K)).to(dev) for _ in range(4)]
        b = torch.randn(N, generator=g).to(dev)
        R = torch.randn(M, N, generator=g).to(dev) if epi == 2 else None
        i = [0]

        def run():
            ops.gemm(A, Ws[i[0] % 4], b, epi, R); i[0] += 1
        us = min(ev(run, 40) for _ in range(3))
        pts.append((K, us))
    n = len(pts); sx = sum(k for k, _ in pts); sy = sum(u for _, u in pts)
    sxx = sum(k * k for k, _ in pts); sxy = sum(k * u for k, u in pts)
    slope = (n * sxy - sx * sy) / (n * sxx - sx * sx); t0 = (sy - slope * sx) / n
    print(f"M={M} N={N} epi={epi}: " + "  ".join(f"K={k}: {u:.1f}" for k, u in pts) +
          f"   -> fixed {t0:.1f} us, K loop {2.0 * M * N / slope / 1e6:.1f} TF", flush=True)
