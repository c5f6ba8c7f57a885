"""One split-bf16 GEMM shape in a loop (profiling target): python tools/x3_one.py [M N K epi reps]"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import _tunelib  # noqa: F401  (PIPS_LIB_PATH -> pips_amd._lib.use_library)
from pips_amd import ops
M, N, K, epi, reps = (int(v) for v in sys.argv[1:6]) if len(sys.argv) >= 6 else (2048, 2048, 512, 1, 50)
g = torch.Generator().manual_seed(0)
A = torch.randn(M, K, generator=g).cuda()
W = (torch.randn(N, K, generator=g) / math.sqrt(K)).cuda()
b = torch.randn(N, generator=g).cuda()
R = torch.randn(M, N, generator=g).cuda() if epi == 2 else None
W3 = ops.split_bf16x3(W)
for _ in range(reps):
    ops.gemm_x3(A, W3, b, epi, R)
torch.cuda.synchronize()
