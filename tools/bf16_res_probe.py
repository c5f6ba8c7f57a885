"""Per-super-stage cost and fixed cost of gemm_bf16_res_asm_kernel: M = 16384, N = 512 at K = 256 .. 4096 (64 K values per super-stage)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import _tunelib  # noqa: F401  (PIPS_LIB_PATH -> pips_amd._lib.use_library)
from pips_amd import _lib
lib = _lib.load()
dev = "cuda:0"
M, N = 16384, 512
g = torch.Generator().manual_seed(0)
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
def t(fn, n=40):
    for _ in range(8): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
b = torch.randn(N, generator=g).to(dev)
R = torch.randn(M, N, generator=g).to(dev)
Cm = torch.empty(M, N, device=dev)
for K in (256, 512, 1024, 2048, 4096):
    A = torch.randn(M, K, generator=g).to(dev).bfloat16()
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev).bfloat16()
    f = lambda: lib.pips_gemm_bf16(_lib.ptr(A), 1, K, _lib.ptr(W), _lib.ptr(b), _lib.ptr(Cm), 0, N, M, N, K, 2, _lib.ptr(R), N, st())
    assert f() == 0
    us = t(f)
    print(f"K={K:5d} ({K//64:3d} super-stages): {us:6.1f} us  route {lib.pips_gemm_bf16_route(M, N, K, 2, 1, 0)}  {2.0*M*N*K/us/1e6:6.0f} TF")
