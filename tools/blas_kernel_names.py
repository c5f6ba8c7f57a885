"""Which library kernels torch.matmul runs for the channel-mix shapes (run under rocprofv3 --kernel-trace --stats): names encode
the macro tile / split-K choices of hipBLASLt / rocBLAS -- reconnaissance, not product code."""
import torch
dev = "cuda:0"
for dt, Ms in ((torch.bfloat16, (16384,)), (torch.float32, (2048, 131072))):
    for M in Ms:
        for (N, K) in ((2048, 512), (512, 2048)):
            a = torch.randn(M, K, device=dev).to(dt); w = torch.randn(N, K, device=dev).to(dt)
            for _ in range(20):
                torch.matmul(a, w.t())
torch.cuda.synchronize()
