"""CPU check of the arithmetic behind the split-bf16 matrix path (pips_amd/csrc/gemm_x3.hip): the
3-way round-to-nearest split is exact, every term is a bf16, and six bf16 products per fp32 product
reproduce an fp32 GEMM to better than fp32 accumulation noise.  numpy restatement of split3_pair();
the GPU tests (test_kernels_gpu.py::test_split_bf16x3_is_exact / test_gemm_split_bf16) check the kernel."""
import numpy as np


def bf16_rne(x):
    """fp32 -> nearest bf16 (ties to even), returned as fp32: what v_cvt_pk_bf16_f32 does."""
    u = np.asarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


def split3(x):
    x = np.asarray(x, dtype=np.float32)
    h = bf16_rne(x)
    r = (x - h).astype(np.float32)
    m = bf16_rne(r)
    l = (r - m).astype(np.float32)
    return h, m, l


def test_split_is_exact_and_terms_are_bf16():
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.standard_normal(20000), rng.standard_normal(20000) * 1e-5,
                        rng.standard_normal(20000) * 1e5, [0.0, 1.0, -1.0, 65504.0, 3e-30]]).astype(np.float32)
    h, m, l = split3(x)
    assert np.array_equal((h.astype(np.float64) + m + l).astype(np.float32), x)
    for t in (h, m, l):
        assert not np.any(t.view(np.uint32) & np.uint32(0xFFFF)), "a term has bits below bf16 precision"
    nz = x != 0
    assert np.all(np.abs(m[nz]) <= np.abs(x[nz]) * 2.0 ** -8) and np.all(np.abs(l[nz]) <= np.abs(x[nz]) * 2.0 ** -16)


def test_six_products_are_fp32_grade():
    rng = np.random.default_rng(1)
    M, K, N = 64, 2048, 96
    A = rng.standard_normal((M, K)).astype(np.float32)
    W = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    ref = A.astype(np.float64) @ W.astype(np.float64).T
    Ah, Am, Al = (t.astype(np.float64) for t in split3(A))
    Wh, Wm, Wl = (t.astype(np.float64) for t in split3(W))
    # exact products, wide accumulation: isolates what the split drops (am*bl + al*bm + al*bl)
    x3 = Al @ Wh.T + Ah @ Wl.T + Am @ Wm.T + Am @ Wh.T + Ah @ Wm.T + Ah @ Wh.T
    f32 = (A @ W.T).astype(np.float64)                       # an fp32 GEMM's own rounding noise
    scale = np.abs(ref).max()
    assert np.abs(x3 - ref).max() < 2.0 ** -20 * scale
    assert np.abs(x3 - ref).max() < 0.05 * np.abs(f32 - ref).max()
    x1 = Ah @ Wh.T                                           # plain bf16 operands for contrast
    assert np.abs(x1 - ref).max() > 1000 * np.abs(x3 - ref).max()
