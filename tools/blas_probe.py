"""What stock rocBLAS/hipBLASLt (torch.matmul, fp32) reaches on the two channel-mix shapes: a yardstick, not product code."""
import torch, time
dev = "cuda:0"
def t(fn, n=50):
    for _ in range(5): fn()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for M in (2048, 16384, 131072):
    for (N, K) in ((2048, 512), (512, 2048)):
        a = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev)
        us = t(lambda: torch.matmul(a, w.t()))
        print(f"M={M:6d} N={N:4d} K={K:4d}: {us:8.1f} us  {2.0*M*N*K/us/1e6:6.1f} TFLOP/s")
