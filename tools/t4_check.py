"""gemm_bf16_t4_res_kernel through pips_gemm_bf16 (library of PIPS_LIB_PATH, env hooks of the tuning build): parity against torch on
three shapes, then time per launch at the config-3 down-projection shape."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import _tunelib  # noqa: F401
from pips_amd import _lib
lib = _lib.load()
dev = "cuda:0"
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
tag = os.path.basename(os.environ.get("PIPS_LIB_PATH", "product")) + " DB=" + os.environ.get("PIPS_BF16_T4_DB", "-")


def run(M, N, K, seed=0):
    g = torch.Generator().manual_seed(seed)
    A = torch.randn(M, K, generator=g).to(dev).bfloat16()
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev).bfloat16()
    b = torch.randn(N, generator=g).to(dev)
    R = torch.randn(M, N, generator=g).to(dev)
    C = torch.empty(M, N, device=dev)
    f = lambda: lib.pips_gemm_bf16(_lib.ptr(A), 1, K, _lib.ptr(W), _lib.ptr(b), _lib.ptr(C), 0, N, M, N, K, 2, _lib.ptr(R), N, st())
    assert f() == 0
    torch.cuda.synchronize()
    ref = A.float() @ W.float().t() + b + R
    return float((C - ref).abs().max()), f


for shp in ((16384, 512, 2048), (32896, 256, 128), (16384, 768, 384)):
    err, _ = run(*shp)
    print(f"[{tag}] {shp}: route {lib.pips_gemm_bf16_route(*shp, 2, 1, 0)} max |err| {err:.2e}", flush=True)
    assert err < 5e-3
for K in (2048, 4096):
    _, f = run(16384, 512, K)
    for _ in range(10): f()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): f()
    e1.record(); e1.synchronize()
    print(f"[{tag}] K={K}: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us", flush=True)
