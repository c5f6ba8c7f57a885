"""Do the config-3 channel-mix GEMMs suffer from power-of-two row strides?  Times the raw C entry pips_gemm_bf16 with the A
operand (and the bf16 output of the up-projection) in a buffer whose row stride is padded by `pad` elements."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import _tunelib  # noqa: F401  (PIPS_LIB_PATH -> pips_amd._lib.use_library)
from pips_amd import _lib
lib = _lib.load()
dev = "cuda:0"
M = 16384
g = torch.Generator().manual_seed(0)
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for pad in (0, 64, 128, 192):
    # down-projection: A = h (M, 2048) bf16 with row stride 2048 + pad, W (512, 2048), C = x (M, 512) fp32 += residual
    K, N = 2048, 512
    lda = K + pad
    Abuf = torch.randn(M, lda, generator=g).to(dev).bfloat16()
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev).bfloat16()
    b = torch.randn(N, generator=g).to(dev)
    R = torch.randn(M, N, generator=g).to(dev)
    Cm = torch.empty(M, N, device=dev)
    f_down = lambda: lib.pips_gemm_bf16(_lib.ptr(Abuf), 1, lda, _lib.ptr(W), _lib.ptr(b), _lib.ptr(Cm), 0, N, M, N, K, 2, _lib.ptr(R), N, st())
    assert f_down() == 0
    # up-projection: A = xn (M, 512) bf16, W (2048, 512), C = h bf16 with row stride 2048 + pad
    K2, N2 = 512, 2048
    ldc = N2 + pad
    A2 = torch.randn(M, K2, generator=g).to(dev).bfloat16()
    W2 = (torch.randn(N2, K2, generator=g) / K2 ** 0.5).to(dev).bfloat16()
    b2 = torch.randn(N2, generator=g).to(dev)
    C2 = torch.empty(M, ldc, device=dev, dtype=torch.bfloat16)
    f_up = lambda: lib.pips_gemm_bf16(_lib.ptr(A2), 1, K2, _lib.ptr(W2), _lib.ptr(b2), _lib.ptr(C2), 1, ldc, M, N2, K2, 1, None, 0, st())
    assert f_up() == 0
    print(f"row pad {pad:3d}: down-projection {t(f_down):6.1f} us (route {lib.pips_gemm_bf16_route(M, N, K, 2, 1, 0)})   up-projection {t(f_up):6.1f} us")
