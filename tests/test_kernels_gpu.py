"""Stage-by-stage parity of the HIP kernels (through the C ABI) against the CPU oracle.

Tolerances: every stage is fp32 arithmetic that differs from the reference only by
summation order, so stage outputs are compared at a few fp32 ulps of the stage's value
scale (stated per test)."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _oracle():
    from oracle import pips_oracle as O
    return O


def _cus():
    """compute units of the device: the route thresholds ("tiles per compute unit") are relative to it"""
    from pips_amd import _lib
    return _lib.load().pips_device_cus()


def _rel_err(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


# ----------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize("M,N,K,epi", [
    (2048, 2048, 512, 1),      # channel-mix up-projection + GELU (128x128 tiles)
    (2048, 512, 2048, 2),      # channel-mix down-projection + residual (64x64 tiles)
    (2048, 512, 544, 0),       # input projection
    (256, 1040, 512, 0),       # head: N not a tile multiple
    (77, 130, 64, 0),          # ragged M and N
    (8, 32, 32, 1),            # tiny
    (16384, 2048, 512, 1),     # config-3 per-GPU rows
    (4096, 2048, 512, 1),      # gemm_f32_t4u_kernel, two row tiles per block
    (16384, 512, 2048, 2),     # the same with the residual epilogue
    (2048, 2048, 128, 1),      # the same, four stages only (no pass through its K loop)
    (3072, 512, 1024, 2),      # gemm_f32_t4d_kernel (64x64 tiles, K split over the waves), 8 stages
    (2048, 384, 512, 2),       # the same at its shortest K, 192 tiles
    (2176, 2048, 576, 1),      # 17 x 16 tiles of 128 (more blocks than compute units, a 1 x 8 XCD split), 18 stages
    (2112, 512, 768, 2),       # 33 x 8 tiles of 64, 24 stages
    (2048, 2176, 512, 2),      # residual form on 128 x 128 tiles with ldc = ldr = 2176
])
def test_gemm_f32(M, N, K, epi):
    from pips_amd import ops
    g = torch.Generator().manual_seed(M * 7 + N)
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) / math.sqrt(K)
    b = torch.randn(N, generator=g)
    R = torch.randn(M, N, generator=g) if epi == 2 else None
    ref = A.double() @ W.double().t() + b.double()
    if epi == 1:
        ref = F.gelu(ref)
    elif epi == 2:
        ref = ref + R.double()
    out = ops.gemm(A.to(DEV), W.to(DEV), b.to(DEV), epi, None if R is None else R.to(DEV)).cpu()
    # asymmetric operands: a transposed C write would fail this by O(1)
    assert _rel_err(out.double(), ref) < 2e-6


def test_gemm_f32_assembly_routes_and_bitwise():
    """The headline's channel-mix Linears reach the four-wave assembly kernels (gemm_f32_t4.hip); the 128 x 128 form keeps
    igemm_f32_kernel's K order, so its result is BITWISE that kernel's (reached here through a null bias)."""
    from pips_amd import ops, _lib
    lib = _lib.load()
    assert lib.pips_gemm_f32_route(2048, 2048, 512, 1) == 1
    assert lib.pips_gemm_f32_route(2048, 512, 2048, 2) == 2
    assert lib.pips_gemm_f32_route(16384, 512, 2048, 2) == 1
    assert lib.pips_gemm_f32_route(2048, 512, 544, 0) == 0        # plain bias epilogue, K % 64 != 0
    assert lib.pips_gemm_f32_route(1024, 2048, 512, 1) == 0       # half a tile per compute unit
    assert lib.pips_gemm_f32_route(256, 1040, 512, 0) == 0
    g = torch.Generator().manual_seed(5)
    for (M, N, K) in ((2048, 2048, 512), (4096, 2048, 512)):
        A = torch.randn(M, K, generator=g).to(DEV)
        W = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(DEV)
        t4 = ops.gemm(A, W, torch.zeros(N, device=DEV), 1)
        ref = ops.gemm(A, W, None, 1)
        assert torch.equal(t4, ref)


def test_gemm_identity_asymmetric():
    """A = I against an asymmetric W catches row/column swaps in the MFMA C/D map."""
    from pips_amd import ops
    K = 64
    A = torch.eye(K)
    W = torch.arange(96 * K, dtype=torch.float32).reshape(96, K) * 1e-3
    out = ops.gemm(A.to(DEV), W.to(DEV)).cpu()
    assert torch.equal(out, W.t().contiguous())


# ----------------------------------------------------------------------------- split-bf16 path
def test_split_bf16x3_is_exact():
    """x = h + m + l exactly, each term a bf16: h = bf16(x), m = bf16(x - h), l the rest.  (Below
    ~2^-110 the remainders are fp32 subnormals and flush to zero: h alone, 8 bits, is kept.)"""
    from pips_amd import ops
    g = torch.Generator().manual_seed(5)
    x = torch.cat([torch.randn(4096, generator=g), torch.randn(4096, generator=g) * 1e-6,
                   torch.randn(4096, generator=g) * 1e6, torch.tensor([0.0, -0.0, 1.0, -1.0, 3.0e-30, 65504.0])])
    x = x[: x.numel() // 2 * 2].contiguous()
    planes = ops.split_bf16x3(x.to(DEV)).cpu()                        # int16 (3, n)
    terms = (planes.to(torch.int32) << 16).view(torch.float32)        # bf16 bits -> fp32 values
    assert torch.equal(terms.double().sum(dim=0).float(), x)
    assert torch.equal(terms[0], x.bfloat16().float())                                 # h = round-to-nearest-even bf16


@pytest.mark.parametrize("M,N,K,epi", [
    (2048, 2048, 512, 1), (2048, 512, 2048, 2), (2048, 512, 544, 0), (256, 1040, 512, 0),
    (77, 132, 96, 2), (8, 32, 32, 1), (16384, 2048, 512, 1),
    (16500, 1040, 512, 0),      # 256x128 tiles with ragged M and N edges
    (16384, 512, 2048, 2),      # the down-projection form at config-3 rows
    (4096, 2048, 128, 1),       # short K
])
def test_gemm_split_bf16(M, N, K, epi):
    """Same reference and the SAME tolerance as test_gemm_f32: the split path is fp32-grade."""
    from pips_amd import ops
    g = torch.Generator().manual_seed(M * 7 + N)
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) / math.sqrt(K)
    b = torch.randn(N, generator=g)
    R = torch.randn(M, N, generator=g) if epi == 2 else None
    ref = A.double() @ W.double().t() + b.double()
    if epi == 1:
        ref = F.gelu(ref)
    elif epi == 2:
        ref = ref + R.double()
    W3 = ops.split_bf16x3(W.to(DEV))
    out = ops.gemm_x3(A.to(DEV), W3, b.to(DEV), epi, None if R is None else R.to(DEV)).cpu()
    assert _rel_err(out.double(), ref) < 2e-6
    exact = ops.gemm(A.to(DEV), W.to(DEV), b.to(DEV), epi, None if R is None else R.to(DEV)).cpu()
    # as close to fp64 as the exact-fp32 MFMA kernels.  Both results carry only fp32 rounding noise and the MAXIMUM over millions of
    # outputs of two different summation orders fluctuates by tens of percent (3.5e-6 against 2.7e-6 at K = 128): slack 1.5x; 2x where
    # the exact kernel is gemm_f32_t4d_kernel (M = 2048 down-projection: four independent partial sums per output, 2.9e-6 against 4.9e-6)
    from pips_amd import _lib
    slack = 2.0 if _lib.load().pips_gemm_f32_route(M, N, K, epi) == 2 else 1.5
    assert float((out.double() - ref).abs().max()) <= slack * float((exact.double() - ref).abs().max()) + 1e-7


@pytest.mark.parametrize("F_,H,W,Cin,Cout,k,s,p", [
    (2, 46, 62, 64, 64, 3, 1, 1), (2, 46, 62, 64, 96, 3, 2, 1), (2, 23, 31, 96, 128, 1, 2, 0),
    (1, 16, 20, 416, 256, 3, 1, 1), (8, 92, 124, 64, 64, 3, 1, 1), (8, 46, 62, 128, 128, 3, 1, 1),
])
def test_conv_split_bf16(F_, H, W, Cin, Cout, k, s, p):
    from pips_amd import ops
    g = torch.Generator().manual_seed(Cin + Cout + k)
    x = torch.randn(F_, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(Cin * k * k)
    b = torch.randn(Cout, generator=g)
    ref = F.conv2d(x.double(), w.double(), b.double(), stride=s, padding=p).permute(0, 2, 3, 1)
    w3 = ops.split_bf16x3(w.permute(0, 2, 3, 1).contiguous().to(DEV))
    out, stats = ops.conv_nhwc_x3(x.permute(0, 2, 3, 1).contiguous().to(DEV), w3, b.to(DEV), k, s, p, want_stats=True)
    assert _rel_err(out.cpu().double(), ref) < 2e-6
    s1, s2 = ops.partial_sums(stats.cpu())
    assert _rel_err(s1, ref.sum(dim=(1, 2))) < 1e-5
    assert _rel_err(s2, (ref * ref).sum(dim=(1, 2))) < 1e-5


@pytest.mark.parametrize("F_,H,W,Cin,Cout,k,s,p", [
    (8, 92, 124, 96, 96, 3, 1, 1),       # 360 tiles of 256x128: the 8-wave conv tile, Cout = 96 in a 128-wide tile
    (8, 46, 62, 416, 256, 3, 1, 1),      # 192 tiles, K = 3744
    (8, 184, 248, 64, 96, 3, 2, 1),      # stride 2, ragged last m-tile (11408 = 44 * 256 + 144)
])
def test_conv_split_bf16_256_row_tile(F_, H, W, Cin, Cout, k, s, p):
    """Config-2 layer shapes that take the 256-row split tile, against the exact-fp32 MFMA conv
    (itself checked against fp64 above): outputs and instance-norm partial sums."""
    from pips_amd import ops
    g = torch.Generator().manual_seed(Cin * 3 + Cout)
    x = torch.randn(F_, H, W, Cin, generator=g).to(DEV)
    w = (torch.randn(Cout, k, k, Cin, generator=g) / math.sqrt(Cin * k * k)).to(DEV)
    b = torch.randn(Cout, generator=g).to(DEV)
    ref, rst = ops.conv_nhwc(x, w, b, k, s, p, want_stats=True)
    out, st = ops.conv_nhwc_x3(x, ops.split_bf16x3(w), b, k, s, p, want_stats=True)
    assert st.shape[1] == 4 * ((ref.shape[1] * ref.shape[2] + 255) // 256)    # the 256-row tile (4 wave rows) was taken
    assert _rel_err(out.double(), ref.double()) < 4e-6
    (a1, a2), (b1, b2) = ops.partial_sums(st.cpu()), ops.partial_sums(rst.cpu())
    assert _rel_err(a1, b1) < 1e-5 and _rel_err(a2, b2) < 1e-5


# ----------------------------------------------------------------------------- conv
@pytest.mark.parametrize("F_,H,W,Cin,Cout,k,s,p", [
    (2, 46, 62, 64, 64, 3, 1, 1),
    (2, 46, 62, 64, 96, 3, 2, 1),
    (1, 23, 31, 96, 96, 3, 1, 1),
    (2, 23, 31, 96, 128, 1, 2, 0),
    (1, 16, 20, 416, 256, 3, 1, 1),
    (1, 16, 20, 256, 128, 1, 1, 0),
    (8, 92, 124, 64, 64, 3, 1, 1),     # enough blocks for the 128-row tile
    (16, 92, 124, 64, 64, 3, 1, 1),    # >= 2.5 tiles per compute unit: the four-wave assembly kernels of conv_f32_t4.hip (256-pixel tiles)
    (16, 93, 125, 64, 64, 3, 1, 1),    # the same, odd size: ragged last tile, column flags at every phase of a piece
    (400, 20, 13, 64, 64, 3, 1, 1),    # the same on tiny frames (two tiles each, the second nearly empty; image rows shorter than a piece run)
    (8, 92, 124, 96, 96, 3, 1, 1),     # 96 -> 96: 128-pixel tiles, three channel blocks
    (8, 93, 125, 96, 96, 3, 1, 1),
    (8, 46, 62, 416, 256, 3, 1, 1),    # conv2: four column tiles of 64 channels, 117 stages per tile
    (8, 45, 63, 416, 256, 3, 1, 1),
    (640, 5, 7, 64, 64, 3, 1, 1),      # one ragged tile per frame, seven-pixel image rows: every piece straddles rows, most taps fall outside
    (1300, 3, 3, 96, 96, 3, 1, 1),     # the smallest frame the kernels take
    (3, 300, 401, 64, 64, 3, 1, 1),    # few large frames (frames % 8 != 0: the linear block order)
])
def test_conv_nhwc(F_, H, W, Cin, Cout, k, s, p):
    from pips_amd import ops
    g = torch.Generator().manual_seed(Cin + Cout + k)
    x = torch.randn(F_, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(Cin * k * k)
    b = torch.randn(Cout, generator=g)
    ref = F.conv2d(x.double(), w.double(), b.double(), stride=s, padding=p).permute(0, 2, 3, 1)
    wp = w.permute(0, 2, 3, 1).contiguous()
    out, stats = ops.conv_nhwc(x.permute(0, 2, 3, 1).contiguous().to(DEV), wp.to(DEV), b.to(DEV), k, s, p,
                               want_stats=True)
    out = out.cpu()
    # fp32 accumulation over K = k*k*Cin terms (igemm_f32_kernel's order in every kernel); conv2's 3744-term sums at 5.8 M outputs reach 2.0e-6
    assert _rel_err(out.double(), ref) < (2e-6 if k * k * Cin < 2000 else 4e-6)
    px, ncol = (256, 1) if Cin == 64 else (128, 4 if Cin == 416 else 1)
    if k == 3 and s == 1 and (Cin, Cout) in ((64, 64), (96, 96), (416, 256)) and F_ * ncol * ((H * W + px - 1) // px) * 2 >= 5 * _cus():
        # conv_f32_t4.hip was taken (>= 2.5 tiles per compute unit): one partial per wave (px / 4 pixels)
        assert stats.shape[1] == (H * W + px // 4 - 1) // (px // 4)
    s1, s2 = ops.partial_sums(stats.cpu())                     # (F, Cout) each
    assert _rel_err(s1, ref.sum(dim=(1, 2))) < 1e-5
    assert _rel_err(s2, (ref * ref).sum(dim=(1, 2))) < 1e-5


@pytest.mark.parametrize("F_,H,W,Cin,Cout", [
    (16, 92, 124, 64, 64),       # 23 x 2 tiles per frame, the last column tile 60 wide: the LDS-resident 3x3 kernel (conv_bf16_c64.hip)
    (4, 186, 250, 64, 64),       # rows and columns ragged (186 = 46*4 + 2, 250 = 3*64 + 58)
    (2, 46, 62, 64, 64),         # too few tiles: the implicit-GEMM bf16 kernel, same contract
    (8, 92, 124, 96, 96),        # other channel counts: implicit GEMM
])
def test_conv_nhwc_bf16(F_, H, W, Cin, Cout):
    """3x3 convolution with bf16 MFMA operands against conv2d of the bf16-rounded operands in fp64, and its InstanceNorm
    partials against the sums of the output it wrote."""
    from pips_amd import ops
    g = torch.Generator().manual_seed(F_ + H + Cin)
    x = torch.randn(F_, H, W, Cin, generator=g)
    w = torch.randn(Cout, 3, 3, Cin, generator=g) / math.sqrt(Cin * 9)
    b = torch.randn(Cout, generator=g)
    out, stats = ops.conv_nhwc_bf16(x.to(DEV), w.to(DEV).bfloat16(), b.to(DEV), 3, 1, 1, want_stats=True)
    out = out.cpu()
    xr, wr = x.bfloat16().double(), w.bfloat16().double()
    ref = F.conv2d(xr.permute(0, 3, 1, 2), wr.permute(0, 3, 1, 2), b.double(), stride=1, padding=1).permute(0, 2, 3, 1)
    assert _rel_err(out.double(), ref) < 2e-6
    s1, s2 = ops.partial_sums(stats.cpu())
    assert _rel_err(s1, out.double().sum(dim=(1, 2))) < 1e-5
    assert _rel_err(s2, (out.double() ** 2).sum(dim=(1, 2))) < 1e-5


@pytest.mark.parametrize("F_,H,W,Cin,Cout,k,s,norm,out_bf16", [
    (16, 92, 124, 64, 64, 3, 1, True, True),     # LDS-resident 64 -> 64 kernel, producer's InstanceNorm + ReLU applied on load
    (4, 186, 250, 64, 64, 3, 1, True, True),     # ragged rows and columns
    (16, 92, 124, 64, 64, 3, 1, False, True),    # the same kernel on a finished activation
    (2, 46, 62, 64, 64, 3, 1, False, True),      # too few tiles: implicit GEMM on bf16 maps
    (8, 92, 124, 96, 96, 3, 1, False, True),     # Cout = 96 on a 128-wide tile
    (24, 92, 124, 96, 96, 3, 1, False, True),    # >= 4 tiles of 256 pixels per compute unit: the four-wave kernel of conv_bf16_t4c.hip
    (24, 93, 125, 96, 96, 3, 1, False, True),    # the same, odd size: ragged last tile, column flags at every phase of a piece
    (64, 20, 13, 96, 96, 3, 1, False, True),     # the same on tiny frames (two tiles each, image rows shorter than a piece run)
    (8, 93, 125, 64, 96, 3, 2, False, True),     # stride 2, odd size
    (8, 92, 124, 64, 96, 1, 2, False, True),     # the 1x1 stride-2 shortcut
    (8, 46, 62, 256, 128, 1, 1, False, False),   # conv3: bf16 map in, fp32 pyramid out
    (24, 46, 62, 416, 256, 3, 1, False, True),   # conv2 with enough frames for the 128 x 256 tile (ragged last row tile)
    (4, 46, 62, 416, 256, 3, 1, False, True),    # conv2 below that: 128-wide tiles
])
def test_conv_nhwc_bf16_maps(F_, H, W, Cin, Cout, k, s, norm, out_bf16):
    """Convolutions of the bf16 encoder mode: bf16 map in (optionally normalised + ReLU'd while staged), bf16 or fp32 map out,
    against conv2d in fp64 of exactly the operands the kernel is specified to form; statistics against the fp32 values."""
    from pips_amd import ops
    g = torch.Generator().manual_seed(F_ + H + Cin + k)
    x = (torch.randn(F_, H, W, Cin, generator=g) * 1.5 + 0.3).bfloat16()
    w = (torch.randn(Cout, k, k, Cin, generator=g) / math.sqrt(Cin * k * k)).bfloat16()
    b = torch.randn(Cout, generator=g)
    nrm = None
    xin = x.float()
    if norm:
        mean = torch.randn(F_, Cin, generator=g) * 0.3
        rstd = torch.rand(F_, Cin, generator=g) + 0.5
        nrm = torch.stack([mean, rstd], dim=-1)
        xin = torch.relu((xin - mean[:, None, None, :]) * rstd[:, None, None, :]).bfloat16().float()   # one rounding, on staging
    p = k // 2
    out, stats = ops.conv_nhwc_bf16_maps(x.to(DEV), w.to(DEV), b.to(DEV), k, s, p, in_norm=None if nrm is None else nrm.to(DEV),
                                         out_bf16=out_bf16, want_stats=True)
    ref = F.conv2d(xin.double().permute(0, 3, 1, 2), w.double().permute(0, 3, 1, 2), b.double(), stride=s, padding=p).permute(0, 2, 3, 1)
    out = out.cpu()
    assert out.dtype == (torch.bfloat16 if out_bf16 else torch.float32) and tuple(out.shape) == tuple(ref.shape)
    if out_bf16:
        # the stored map is the RNE rounding of the fp32 result: within half a bf16 ulp (+ fp32 summation noise) of fp64
        # (normalise-on-load: the kernel forms x * rstd - mean * rstd, a staged value may round to the neighbouring bf16)
        err = (out.double() - ref).abs()
        assert bool((err <= ref.abs() * 2.0 ** -8 + (3e-3 if norm else 1e-5)).all()), float((err / (ref.abs() + 1e-3)).max())
    else:
        assert _rel_err(out.double(), ref) < 2e-6
    s1, s2 = ops.partial_sums(stats.cpu())
    t4c = out_bf16 and Cin == 96 and Cout == 96 and k == 3 and s == 1 and F_ * ((H * W + 255) // 256) >= 4 * _cus()
    if t4c:
        # conv_bf16_t4c.hip: the statistics are those of the STORED bf16 map (what autocast's instance_norm sees)
        assert _rel_err(s1, out.double().sum(dim=(1, 2))) < 1e-5
        assert _rel_err(s2, (out.double() ** 2).sum(dim=(1, 2))) < 1e-5
        assert _rel_err(s1, ref.sum(dim=(1, 2))) < 5e-3 and _rel_err(s2, (ref * ref).sum(dim=(1, 2))) < 5e-3
    else:
        assert _rel_err(s1, ref.sum(dim=(1, 2))) < 1e-5            # statistics come from the fp32 accumulators
        assert _rel_err(s2, (ref * ref).sum(dim=(1, 2))) < 1e-5


def test_conv_nhwc_bf16_maps_without_bias():
    """pips_conv_nhwc_bf16_maps documents bias as '[N] or null'.  The four-wave 96 -> 96 kernel always reads a bias through a
    buffer descriptor, so a layer without one must stay on the register-staged kernel (conv_c96_t4_takes) -- at the shape
    that otherwise takes the four-wave route, and with a statistics capacity that only suits the generic kernel's partition."""
    from pips_amd import ops
    F_, H, W, Cc = 24, 92, 124, 96
    g = torch.Generator().manual_seed(77)
    x = (torch.randn(F_, H, W, Cc, generator=g) * 1.5 + 0.3).bfloat16()
    w = (torch.randn(Cc, 3, 3, Cc, generator=g) / math.sqrt(Cc * 9)).bfloat16()
    out, stats = ops.conv_nhwc_bf16_maps(x.to(DEV), w.to(DEV), None, 3, 1, 1, out_bf16=True, want_stats=True)
    ref = F.conv2d(x.double().permute(0, 3, 1, 2), w.double().permute(0, 3, 1, 2), None, stride=1, padding=1).permute(0, 2, 3, 1)
    err = (out.cpu().double() - ref).abs()
    assert bool((err <= ref.abs() * 2.0 ** -8 + 1e-5).all()), float((err / (ref.abs() + 1e-3)).max())
    s1, s2 = ops.partial_sums(stats.cpu())
    assert _rel_err(s1, ref.sum(dim=(1, 2))) < 1e-5 and _rel_err(s2, (ref * ref).sum(dim=(1, 2))) < 1e-5   # fp32 accumulators: generic kernel


# ----------------------------------------------------------------------------- encoder
@pytest.mark.parametrize("split", [False, True])
# (106 x 150: a width that is no multiple of 4 -- the stem's row-by-row staging; every other size takes its aligned quad loads)
@pytest.mark.parametrize("H,W,stride", [(128, 160, 8), (96, 128, 4), (136, 200, 8), (368, 496, 8), (106, 150, 8)])
def test_encoder_pyramid(H, W, stride, split, weights_raw, arenas):
    from pips_amd import ops
    O = _oracle()
    g = torch.Generator().manual_seed(3)
    rgbs = torch.randint(0, 256, (8, 3, H, W), generator=g).float()
    taps = {}
    fm = O.encoder(weights_raw, 2 * (rgbs / 255.0) - 1.0, stride, taps)
    pyr_ref = O.build_pyramid(fm.unsqueeze(0))
    pyr = ops.encoder_fwd(arenas["raw"], rgbs.to(DEV), stride, split=split)   # one tolerance for both matrix paths
    torch.cuda.synchronize()
    levels = ops.pyramid_levels(pyr, 8, H, W, stride)
    for l, (got, ref) in enumerate(zip(levels, pyr_ref)):
        ref = ref[0].permute(0, 2, 3, 1)
        assert tuple(got.shape) == tuple(ref.shape)
        err = float((got.cpu() - ref).abs().max())
        # feature maps are O(1); 21 normalised convs deep, fp32 reorder noise stays < 5e-5
        assert err < 2e-4, f"level {l}: {err}"


@pytest.mark.parametrize("kind", ["low_contrast", "letterbox", "flat_with_dot"])
def test_encoder_low_variance_frames(kind, weights_raw, arenas):
    """Frames whose channels have |mean| >> std (flat video, letterboxing): the InstanceNorm statistics of the stem and of
    every conv layer are summed about a pivot and combined in fp64 (sum / sum-of-squares partials lose their digits exactly
    there).  Yardstick: the reference arithmetic's own fp32 error against an fp64 run -- the HIP maps stay within 4x of it
    (measured 1.0x / 1.6x / 0.9x)."""
    from pips_amd import ops
    O = _oracle()
    g = torch.Generator().manual_seed(3)
    H, W = 128, 160
    if kind == "low_contrast":
        rgbs = (200 + torch.randint(-2, 3, (8, 3, H, W), generator=g)).float()
    elif kind == "letterbox":
        rgbs = torch.cat([torch.zeros(8, 3, 40, W), torch.randint(0, 256, (8, 3, 48, W), generator=g).float(),
                          torch.zeros(8, 3, 40, W)], 2)
    else:
        rgbs = torch.full((8, 3, H, W), 128.0)
        rgbs[0, 0, 64, 80] = 255.0
    x = 2 * (rgbs / 255.0) - 1.0
    ref32 = O.encoder(weights_raw, x, 8)
    ref64 = O.encoder(O.to_dtype(weights_raw, torch.float64), x.double(), 8)
    pyr = ops.encoder_fwd(arenas["raw"], rgbs.to(DEV), 8)
    got = ops.pyramid_levels(pyr, 8, H, W, 8)[0].cpu().permute(0, 3, 1, 2)
    floor = float((ref32.double() - ref64).abs().max())
    err = float((got.double() - ref64).abs().max())
    print(f"{kind}: HIP vs fp64 {err:.2e}, fp32 reference arithmetic vs fp64 {floor:.2e}, |map| {float(ref64.abs().max()):.1f}")
    assert err < 4 * floor + 1e-5


# the second size takes the fused LDS-resident layer-1 path; the third has a width that is no multiple of 4 (the stem's pair-wise staging)
@pytest.mark.parametrize("F_,H,W", [(8, 128, 160), (16, 184, 248), (8, 106, 150)])
def test_encoder_bf16_operands(F_, H, W, weights_raw, arenas):
    """bf16 encoder mode (config 3: bf16 conv operands AND bf16 activation maps, the rounding points of the reference under
    torch.autocast(bfloat16)): maps within bf16-level error of the fp32 oracle, and no further from the oracle run under
    autocast than that run is from fp32."""
    from pips_amd import ops
    O = _oracle()
    g = torch.Generator().manual_seed(3)
    rgbs = torch.randint(0, 256, (F_, 3, H, W), generator=g).float()
    x = 2 * (rgbs / 255.0) - 1.0
    fm = O.encoder(weights_raw, x, 8)
    with torch.autocast("cpu", dtype=torch.bfloat16):
        fm_ac = O.encoder(weights_raw, x, 8).float()
    pyr = ops.encoder_fwd(arenas["raw"], rgbs.to(DEV), 8, bf16=True)
    got = ops.pyramid_levels(pyr, F_, H, W, 8)[0].cpu()
    ref, ref_ac = fm.permute(0, 2, 3, 1), fm_ac.permute(0, 2, 3, 1)
    rms = lambda a, b: float(((a - b) ** 2).mean().sqrt() / (b ** 2).mean().sqrt())
    rel = float((got - ref).abs().max() / ref.abs().max())
    print(f"bf16 encoder {F_}x{H}x{W}: vs fp32 oracle max rel {rel:.2e} rms {rms(got, ref):.2e}; vs autocast oracle rms "
          f"{rms(got, ref_ac):.2e}; autocast oracle vs fp32 oracle rms {rms(ref_ac, ref):.2e}")
    assert 1e-5 < rel < 8e-2 and rms(got, ref) < 4e-2          # 22 convs x 2^-9 operand + activation rounding
    assert rms(got, ref_ac) < 1.5 * rms(ref_ac, ref) + 5e-3
    # levels 1-3 are 2x2 means of the fp32 level-0 map
    lv = ops.pyramid_levels(pyr, F_, H, W, 8)
    pooled = torch.nn.functional.avg_pool2d(lv[0].permute(0, 3, 1, 2), 2, stride=2).permute(0, 2, 3, 1)
    assert float((lv[1] - pooled).abs().max()) < 1e-5
    # decoded uint8 frames (PIPS_FLAG_RGB_U8) through the same stem: the same values, bit for bit
    pyr8 = ops.encoder_fwd(arenas["raw"], rgbs.to(torch.uint8).to(DEV), 8, bf16=True)
    for a8, af in zip(ops.pyramid_levels(pyr8, F_, H, W, 8), lv):
        assert torch.equal(a8, af)


# ----------------------------------------------------------------------------- tracker stages
def _random_state(B, N, H8, W8, seed=5, spread=1.0):
    g = torch.Generator().manual_seed(seed)
    S, C = 8, 128
    fmaps = torch.randn(B, S, C, H8, W8, generator=g)
    ffeats = torch.randn(B, S, N, C, generator=g)
    coords = torch.rand(B, S, N, 2, generator=g) * torch.tensor([W8 - 1.0, H8 - 1.0])
    coords = coords + torch.randn(B, S, N, 2, generator=g) * spread
    # out-of-map and border cases
    coords[0, 0, 0] = torch.tensor([-5.0, -7.5])
    coords[0, 1, 0] = torch.tensor([W8 + 6.0, H8 + 2.0])
    coords[0, 2, 0] = torch.tensor([0.0, 0.0])
    coords[0, 3, 0] = torch.tensor([W8 - 1.0, H8 - 1.0])
    coords[0, 4, 0] = torch.tensor([2.0, 3.0])
    coords[0, 5, 0] = torch.tensor([-0.25, H8 - 0.5])
    return fmaps, ffeats, coords


def _pack_pyramid(pyr_ref, B, H, W, stride):
    """oracle pyramid list (B,S,C,h,w) -> packed channel-last device buffer."""
    from pips_amd import _lib
    lib = _lib.load()
    Fr = B * 8
    buf = torch.zeros(lib.pips_pyramid_floats(Fr, H, W, stride), dtype=torch.float32)
    for l, p in enumerate(pyr_ref):
        off = lib.pips_pyramid_offset(Fr, H, W, stride, l)
        flat = p.reshape(Fr, 128, p.shape[-2], p.shape[-1]).permute(0, 2, 3, 1).reshape(-1)
        buf[off:off + flat.numel()] = flat
    return buf.to(DEV)


def _pm(t):
    """(B,S,N,X) -> particle-major (B*N*S, X)."""
    B, S, N, X = t.shape
    return t.permute(0, 2, 1, 3).reshape(B * N * S, X).contiguous()


@pytest.mark.parametrize("B,N,H8,W8,spread", [(1, 300, 46, 62, 0.7), (2, 1024, 33, 40, 3.0), (1, 77, 16, 20, 0.0), (1, 256, 46, 62, 4.0),
                                              (1, 700, 17, 24, 0.5)])
def test_mixer_input_build_tiled_bf16_mfma(B, N, H8, W8, spread):
    """The bf16 mode's tiled gather on the matrix cores (gather_mfma_kernel, PIPS_FLAG_BF16_MAPS): under autocast the reference
    correlates bf16 features with bf16 maps (nets/pips.py:394-397).  Against the oracle's CorrBlock on maps AND features rounded
    to bf16, evaluated in fp64 (the products of two bf16 numbers are exact in fp32, so only the order of the fp32 sums differs:
    fp32 tolerance) -- dense grid in one frame (items of > 96 particles are split), far-out points, windows half outside the map,
    tile / pixel boundaries, a tile with a single particle, maps smaller than a tile."""
    from pips_amd import ops
    O = _oracle()
    fmaps, ffeats, coords = _random_state(B, N, H8, W8, seed=19, spread=spread)
    n_ = int(N ** 0.5)
    gy, gx = torch.meshgrid(torch.linspace(0, H8 - 1, n_), torch.linspace(0, W8 - 1, n_), indexing="ij")
    coords[0, 0, :n_ * n_] = torch.stack([gx.reshape(-1), gy.reshape(-1)], -1)
    coords[0, 6, 1] = torch.tensor([-40.0, 3.0])
    coords[0, 7, 1] = torch.tensor([W8 + 3.5, H8 + 9.0])
    coords[0, 5, 2] = torch.tensor([min(15.999999, W8 - 1.0), min(16.0, H8 - 1.0)])            # on a tile boundary
    coords[0, 5, 3] = torch.tensor([-2.0, 1.5])                                                  # window partly outside
    coords[0, 1, :] = torch.tensor([W8 * 0.5, H8 * 0.5])                                         # a whole frame in ONE cell: items split at 96
    pyr_ref = O.build_pyramid(fmaps)
    pyr_bf = [p.bfloat16().double() for p in pyr_ref]
    ref = O.corr_sample(pyr_bf, ffeats.bfloat16().double(), coords.double())                     # (B,8,N,196)
    pyr = ops.pyramid_mirror(_pack_pyramid(pyr_ref, B, H8 * 8, W8 * 8, 8), B * 8, H8 * 8, W8 * 8, 8)
    ff, co = _pm(ffeats).to(DEV), _pm(coords).to(DEV)
    X = ops.mixer_input_build_tiled(pyr, B, H8, W8, ff, co, bf16_maps=True).cpu()
    Xd = ops.mixer_input_build(pyr, B, H8, W8, ff, co, bf16_maps=True).cpu()                    # direct kernel: the SAME bf16 x bf16 products (round 6)
    assert torch.equal(X[:, :128], Xd[:, :128]) and torch.equal(X[:, 324:], Xd[:, 324:])       # features, embedding, padding
    err = float((X[:, 128:324].double() - _pm(ref)).abs().max())
    errd = float((Xd[:, 128:324].double() - _pm(ref)).abs().max())
    dd = float((X[:, 128:324] - Xd[:, 128:324]).abs().max())
    print(f"bf16 matrix-core gather vs fp64 oracle on bf16 operands {err:.2e}; the direct bf16-map kernel vs the same oracle {errd:.2e}; "
          f"route against route {dd:.2e}")
    # one rounding contract whatever the route: the two kernels differ by the order of their fp32 sums only
    assert err < 1e-4 and errd < 1e-4 and dd < 1e-4          # (sample positions: fp32 here, fp64 in the yardstick)
    assert torch.equal(X.view(B * N, 8, 544)[1, 6, 128:324], torch.zeros(196))                   # fully outside the map: zeros padding


@pytest.mark.parametrize("B,N,H8,W8", [(1, 33, 46, 62), (2, 7, 17, 25)])
def test_score_map_terms(B, N, H8, W8):
    """The dense score maps of nets/pips.py:501-511 (four levels upsampled with align_corners=True and summed) and the two
    sums score_map_loss (:58-92) takes from each: HIP (levels summed once, heat maps never stored) against the oracle's
    (B,S,N,H8,W8) volume, incl. odd level sizes and targets on the map border."""
    from pips_amd import ops, _lib
    O = _oracle()
    lib = _lib.load()
    fmaps, ffeats, _ = _random_state(B, N, H8, W8, seed=13)
    g = torch.Generator().manual_seed(14)
    tx = torch.randint(0, W8, (B, 8, N), generator=g).float()
    ty = torch.randint(0, H8, (B, 8, N), generator=g).float()
    tx[0, 0, 0], ty[0, 0, 0] = 0.0, 0.0
    tx[0, 1, 0], ty[0, 1, 0] = W8 - 1.0, H8 - 1.0
    use = (torch.rand(B, 8, N, generator=g) > 0.25).float()
    pyr_ref = O.build_pyramid(fmaps)
    fcp = O.dense_score_maps(pyr_ref, ffeats)                                        # (B,S,N,H8,W8)
    a_pos = -fcp[torch.arange(B)[:, None, None], torch.arange(8)[None, :, None], torch.arange(N)[None, None, :],
                 ty.long(), tx.long()]
    sp = lambda a: torch.relu(a) + torch.log(torch.exp(-torch.relu(a)) + torch.exp(a - torch.relu(a)))
    pos_ref = sp(a_pos)
    neg_ref = sp(fcp).sum(dim=(-1, -2)) - sp(-a_pos)
    buf = torch.zeros(lib.pips_pyramid_floats(B * 8, H8 * 8, W8 * 8, 8))
    for l, p in enumerate(pyr_ref):
        off = lib.pips_pyramid_offset(B * 8, H8 * 8, W8 * 8, 8, l)
        flat = p.reshape(B * 8, 128, p.shape[-2], p.shape[-1]).permute(0, 2, 3, 1).reshape(-1)
        buf[off:off + flat.numel()] = flat
    tgt = _pm(torch.stack([tx, ty, use], dim=-1))
    out = ops.score_map_terms(buf.to(DEV), B, H8, W8, _pm(ffeats).to(DEV), tgt.to(DEV)).cpu()
    u = _pm(use.unsqueeze(-1))[:, 0] > 0
    assert torch.equal(out[~u], torch.zeros_like(out[~u]))
    pos_pm, neg_pm = _pm(pos_ref.unsqueeze(-1))[:, 0], _pm(neg_ref.unsqueeze(-1))[:, 0]
    assert float((out[u, 0] - pos_pm[u]).abs().max()) < 1e-4 * max(1.0, float(pos_pm.abs().max()))
    assert float(((out[u, 1] - neg_pm[u]) / neg_pm[u]).abs().max()) < 2e-5


def test_point_sample():
    from pips_amd import ops
    O = _oracle()
    B, N, H8, W8 = 2, 33, 16, 20
    fmaps, _, coords = _random_state(B, N, H8, W8)
    xy = coords[:, 0]                                                # includes (-5,-7.5): clamped indices
    ref = O.point_sample(fmaps[:, 0], xy[..., 0], xy[..., 1])
    lvl0 = fmaps.reshape(B * 8, 128, H8, W8).permute(0, 2, 3, 1).contiguous().to(DEV)
    out = ops.point_sample(lvl0, B, xy.to(DEV)).cpu()
    assert float((out - ref).abs().max()) < 1e-5


@pytest.mark.parametrize("B,N,H8,W8", [(1, 24, 16, 20), (2, 9, 17, 25), (1, 5, 46, 62)])
def test_mixer_input_build(B, N, H8, W8):
    """Fused correlation gather + embedding vs CorrBlock.corr/sample + get_3d_embedding."""
    from pips_amd import ops
    O = _oracle()
    fmaps, ffeats, coords = _random_state(B, N, H8, W8)
    pyr_ref = O.build_pyramid(fmaps)
    X_ref = O.mixer_input(ffeats, O.corr_sample(pyr_ref, ffeats, coords), coords)      # (B*N, S, 519)
    pyr = _pack_pyramid(pyr_ref, B, H8 * 8, W8 * 8, 8)
    X = ops.mixer_input_build(pyr, B, H8, W8, _pm(ffeats).to(DEV), _pm(coords).to(DEV)).cpu()
    X = X.view(B * N, 8, 544)
    assert torch.equal(X[..., 519:], torch.zeros_like(X[..., 519:]))
    assert torch.equal(X[..., :128], X_ref[..., :128])                                  # feature copy
    # correlations are O(|f|^2/sqrt(C)) ~ 11 with 128-term fp32 dots: 2e-5 abs
    err_corr = float((X[..., 128:324] - X_ref[..., 128:324]).abs().max())
    assert err_corr < 5e-5, err_corr
    # sin/cos of arguments up to ~1e4 rad: OCML vs SLEEF differ by <= 2 ulp of 1.0
    err_emb = float((X[..., 324:516] - X_ref[..., 324:516]).abs().max())
    assert err_emb < 5e-6, err_emb
    assert torch.equal(X[..., 516:519], X_ref[..., 516:519])                            # raw flow + time


@pytest.mark.parametrize("B,N,H8,W8", [(1, 24, 16, 20), (2, 9, 17, 25), (1, 5, 46, 62)])
def test_mixer_input_build_bf16_maps(B, N, H8, W8):
    """The gather of the bf16 mode (PIPS_FLAG_BF16_MAPS): the kernel reads the bf16 mirror of the pyramid and rounds the track
    features to bf16 on load -- under autocast torch.matmul casts BOTH operands (nets/pips.py:394-397); products (exact) and sums
    fp32.  Against the oracle's CorrBlock on maps AND features rounded the same way, in fp32 and in fp64 (the yardstick of
    test_config4_gather_bf16_matrix_cores_vs_oracle), border windows included, and within bf16 rounding of the fp32 gather.  The
    feature columns of X stay fp32 (the in-projection rounds them itself)."""
    from pips_amd import ops
    O = _oracle()
    fmaps, ffeats, coords = _random_state(B, N, H8, W8, seed=3)
    coords[0, :, 0] = torch.tensor([-2.0, 1.5])                          # window partly outside the map
    coords[0, :, 1] = torch.tensor([W8 - 1.0, H8 - 1.0])
    pyr_ref = O.build_pyramid(fmaps)
    pyr_bf = [p.bfloat16().float() for p in pyr_ref]                     # the mirror holds bf16(level), level by level
    ff_bf = ffeats.bfloat16().float()
    X_ref = O.mixer_input(ffeats, O.corr_sample(pyr_bf, ff_bf, coords), coords)
    ref64 = O.corr_sample([p.double() for p in pyr_bf], ff_bf.double(), coords.double())          # (B,8,N,196)
    pyr = ops.pyramid_mirror(_pack_pyramid(pyr_ref, B, H8 * 8, W8 * 8, 8), B * 8, H8 * 8, W8 * 8, 8)
    ff, co = _pm(ffeats).to(DEV), _pm(coords).to(DEV)
    X = ops.mixer_input_build(pyr, B, H8, W8, ff, co, bf16_maps=True).cpu().view(B * N, 8, 544)
    X32 = ops.mixer_input_build(pyr, B, H8, W8, ff, co).cpu().view(B * N, 8, 544)
    assert torch.equal(X[..., :128], X_ref[..., :128]) and torch.equal(X[..., 324:], X32[..., 324:])
    err = float((X[..., 128:324] - X_ref[..., 128:324]).abs().max())
    err64 = float((X[..., 128:324].reshape(B * N * 8, 196).double() - _pm(ref64)).abs().max())
    d32 = float((X[..., 128:324] - X32[..., 128:324]).abs().max())
    print(f"bf16-map gather vs oracle on bf16-rounded operands {err:.2e} (fp64 oracle {err64:.2e}); vs the fp32 gather {d32:.2e}")
    assert err < 5e-5 and err64 < 5e-5 and 1e-4 < d32 < 0.3


@pytest.mark.parametrize("B,N,H8,W8,spread", [(1, 300, 46, 62, 0.7), (2, 1024, 33, 40, 3.0), (1, 77, 16, 20, 0.0)])
def test_mixer_input_build_tiled(B, N, H8, W8, spread):
    """LDS-tiled gather (dense query sets) vs the oracle and vs the direct kernel."""
    from pips_amd import ops
    O = _oracle()
    fmaps, ffeats, coords = _random_state(B, N, H8, W8, seed=9, spread=spread)
    # a dense grid in frame 0 of clip 0 (many particles per 16x16 tile) plus far-out points
    n_ = int(N ** 0.5)
    gy, gx = torch.meshgrid(torch.linspace(0, H8 - 1, n_), torch.linspace(0, W8 - 1, n_), indexing="ij")
    coords[0, 0, :n_ * n_] = torch.stack([gx.reshape(-1), gy.reshape(-1)], -1)
    coords[0, 6, 1] = torch.tensor([-40.0, 3.0])
    coords[0, 7, 1] = torch.tensor([W8 + 3.5, H8 + 9.0])
    pyr_ref = O.build_pyramid(fmaps)
    X_ref = O.mixer_input(ffeats, O.corr_sample(pyr_ref, ffeats, coords), coords)
    pyr = _pack_pyramid(pyr_ref, B, H8 * 8, W8 * 8, 8)
    ff, co = _pm(ffeats).to(DEV), _pm(coords).to(DEV)
    X = ops.mixer_input_build_tiled(pyr, B, H8, W8, ff, co).cpu().view(B * N, 8, 544)
    Xd = ops.mixer_input_build(pyr, B, H8, W8, ff, co).cpu().view(B * N, 8, 544)
    assert torch.equal(X[..., :128], X_ref[..., :128]) and torch.equal(X[..., 519:], torch.zeros_like(X[..., 519:]))
    assert float((X[..., 128:324] - X_ref[..., 128:324]).abs().max()) < 5e-5
    assert float((X[..., 128:324] - Xd[..., 128:324]).abs().max()) < 5e-5
    assert torch.equal(X[..., 324:], Xd[..., 324:])                       # embedding: same instructions


@pytest.mark.parametrize("split", [False, True])
@pytest.mark.parametrize("P", [4, 32, 256])
def test_mixer(P, split, weights_raw, arenas):
    from pips_amd import ops
    O = _oracle()
    g = torch.Generator().manual_seed(P)
    x = torch.randn(P, 8, 519, generator=g)
    ref = O.mixer(weights_raw, x)
    X = torch.zeros(P * 8, 544)
    X[:, :519] = x.reshape(P * 8, 519)
    out = ops.mixer_fwd(arenas["raw"], X.to(DEV), split=split).cpu()
    # 12 residual blocks of fp32 GEMMs with K up to 2048; outputs O(1)
    assert float((out - ref).abs().max()) < 1e-4 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("M,N,K,epi,out_bf16", [
    (16384, 2048, 512, 1, True),       # config-3 up-projection: 256 x 256 tiles, two per block (gemm_bf16_t4_gelu_kernel)
    (8192, 2048, 512, 1, True),        # 256 such tiles: one per block (no run-on)
    (6144, 2048, 512, 1, True),        # 192 tiles: three quarters of a tile per compute unit
    (16640, 2048, 512, 1, True),       # 65 row tiles (odd): one tile per block
    (32768, 1024, 512, 1, True),       # 512 tiles on 256 blocks, four column blocks
    (16384, 1920, 512, 1, True),       # N % 256 != 0: the register-staged kernel (the 256 x 128 assembly form of rounds 2-3 is gone)
    (1280, 2048, 512, 1, True),        # below either threshold -> register-staged kernel, same contract
    (16384, 512, 2048, 2, False),      # config-3 down-projection: 128 x 256 tiles, four waves (gemm_bf16_t4_res_kernel)
    (8192, 512, 2048, 2, False),       # the same kernel at half a tile per compute unit (128 tiles)
    (32896, 256, 128, 2, False),       # the same kernel: 257 tiles (not a multiple of the 8 XCDs), two K iterations (the minimum)
    (16384, 768, 192, 2, False),       # three column tiles per row block, three K iterations
    (32768, 384, 1024, 2, False),      # N % 256 != 0: the register-staged kernel
    (4096, 512, 544, 0, False),        # input projection: fp32 A, 32-element K blocks
])
def test_gemm_bf16(M, N, K, epi, out_bf16):
    """pips_gemm_bf16 against torch: bf16 operands, fp32 accumulation, exact GELU, result rounded to the output type.  (The
    four-wave 256 x 256 kernel rounds the Linear's output to bf16 before the GELU, as autocast does; the register-staged one
    feeds the fp32 accumulator to it: one or two bf16 roundings, |gelu'| <= 1.13 -> both within 1.2e-2 of the fp32 GELU of the
    fp32 pre-activation, relative to max(|gelu|, 0.25).)"""
    from pips_amd import ops
    g = torch.Generator().manual_seed(M + N + K)
    a_bf16 = epi != 0
    A = torch.randn(M, K, generator=g).to(DEV)
    A = A.bfloat16() if a_bf16 else A
    W = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(DEV).bfloat16()
    b = torch.randn(N, generator=g).to(DEV)
    R = torch.randn(M, N, generator=g).to(DEV) if epi == 2 else None
    out = ops.gemm_bf16(A, W, b, epi=epi, R=R, out_bf16=out_bf16).float()
    pre = A.bfloat16().float() @ W.float().t() + b
    if epi == 1:
        ref = torch.nn.functional.gelu(pre)
        err = (out - ref).abs() / ref.abs().clamp(min=0.25)
        print(f"gemm_bf16 GELU M={M}: max rel err {float(err.max()):.3e} mean {float(err.mean()):.3e}")
        assert float(err.max()) < 1.2e-2 and float(err.mean()) < 2e-3
    else:
        ref = pre + (R if R is not None else 0)
        assert float((out - ref).abs().max()) < 2e-3 * max(1.0, float(ref.abs().max()))


def test_gemm_bf16_rejects_bf16_residual_flag_without_bf16_output():
    """PIPS_EPI_RES_BF16 (public since ABI 3) is defined beside a bf16 output of the residual epilogue only: with an fp32 C, or on
    another epilogue, the flag used to be ignored silently and a bf16 R read as fp32 -- now PIPS_E_ARG."""
    from pips_amd import ops, _lib
    lib = _lib.load()
    M, N, K = 256, 256, 128
    A = torch.zeros(M, K, device=DEV, dtype=torch.bfloat16)
    W = torch.zeros(N, K, device=DEV, dtype=torch.bfloat16)
    R = torch.zeros(M, N, device=DEV, dtype=torch.bfloat16)
    for out_bf16, epi in ((0, ops.EPI_RESIDUAL | ops.EPI_RES_BF16), (1, ops.EPI_GELU | ops.EPI_RES_BF16), (1, ops.EPI_BIAS | ops.EPI_RES_BF16)):
        Cm = torch.zeros(M, N, device=DEV, dtype=torch.bfloat16 if out_bf16 else torch.float32)
        rc = lib.pips_gemm_bf16(_lib.ptr(A), 1, K, _lib.ptr(W), None, _lib.ptr(Cm), out_bf16, N, M, N, K, epi, _lib.ptr(R), N, ops._stream())
        assert rc != 0 and b"PIPS_EPI_RES_BF16" in lib.pips_last_error()
    assert ops.gemm_bf16(A, W, None, epi=2, R=R, out_bf16=True).dtype == torch.bfloat16          # the defined combination


@pytest.mark.parametrize("M,N,K", [(16384, 512, 2048), (8192, 512, 2048), (32896, 256, 128), (2048, 512, 2048), (1000, 384, 128)])
def test_gemm_bf16_residual_stream(M, N, K):
    """The down-projection with a bf16 residual stream (EPI_RES_BF16: bf16 R, bf16 C -- nets/pips.py:93-100 under autocast): the
    four-wave assembly kernel's second form (first three shapes) and the register-staged kernel (last two, one with ragged tiles)
    against fp32 arithmetic on the same bf16 operands, rounded once: within half a bf16 ulp of the fp64 result (+ fp32 sum noise)."""
    from pips_amd import ops, _lib
    g = torch.Generator().manual_seed(M + N + K + 1)
    A = torch.randn(M, K, generator=g).to(DEV).bfloat16()
    W = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(DEV).bfloat16()
    b = torch.randn(N, generator=g).to(DEV)
    R = (torch.randn(M, N, generator=g) * 3.0).to(DEV).bfloat16()
    out = ops.gemm_bf16(A, W, b, epi=2, R=R, out_bf16=True)
    assert out.dtype == torch.bfloat16
    ref = A.double() @ W.double().t() + b.double() + R.double()
    err = (out.double() - ref).abs()
    assert bool((err <= ref.abs() * 2.0 ** -8 + 2e-5).all()), float((err / (ref.abs() + 1e-2)).max())
    route = _lib.load().pips_gemm_bf16_route(M, N, K, 2 | ops.EPI_RES_BF16, 1, 1)
    takes = M % 128 == 0 and N % 256 == 0 and K % 64 == 0 and (M // 128) * (N // 256) * 2 >= _cus()      # from half a tile per compute unit
    assert route == (3 if takes else 0)


@pytest.mark.parametrize("P", [256, 2048])
def test_mixer_bf16_residual_stream(P, weights_raw, arenas):
    """PIPS_FLAG_BF16_STREAM: the bf16 mixer with its residual stream stored as bf16 (rounded once per residual add).  Against the
    oracle's mixer under torch.autocast(bfloat16) -- whose PreNormResidual holds a bf16 stream too -- it must be no further away
    than the fp32-stream form is, within slack; and it stays at bf16-level distance from the fp32 oracle."""
    from pips_amd import ops
    O = _oracle()
    g = torch.Generator().manual_seed(P + 3)
    x = torch.randn(min(P, 256), 8, 519, generator=g)
    if P > 256:
        x = x.repeat(P // 256, 1, 1) + 0.01 * torch.randn(P, 8, 519, generator=g)
    ref32 = O.mixer(weights_raw, x)
    with torch.autocast("cpu", dtype=torch.bfloat16):
        refbf = O.mixer(weights_raw, x).float()
    X = torch.zeros(P * 8, 544)
    X[:, :519] = x.reshape(P * 8, 519)
    a = ops.mixer_fwd(arenas["raw"], X.to(DEV), bf16=True).cpu()
    b = ops.mixer_fwd(arenas["raw"], X.to(DEV), bf16=True, stream_bf16=True).cpu()
    scale = max(1.0, float(ref32.abs().max()))
    ref32, refbf = ref32.reshape(a.shape), refbf.reshape(a.shape)
    e = lambda u, v: float((u - v).abs().max()) / scale
    print(f"bf16 mixer P={P}: fp32 stream vs autocast oracle {e(a, refbf):.2e}, bf16 stream vs autocast oracle {e(b, refbf):.2e}; "
          f"vs fp32 oracle {e(a, ref32):.2e} / {e(b, ref32):.2e}; autocast oracle vs fp32 oracle {e(refbf, ref32):.2e}; the two forms {e(a, b):.2e}")
    assert torch.isfinite(b).all()
    assert e(b, refbf) < 1.5 * e(a, refbf) + 5e-3 and e(b, ref32) < 3e-2


@pytest.mark.parametrize("P", [32, 256, 2048])
def test_mixer_bf16_operands(P, weights_raw, arenas):
    """bf16 MFMA operands (config 3): against the fp32 oracle at bf16-level tolerance, and much
    closer to an oracle whose big Linear weights/activations are rounded to bf16 the same way."""
    from pips_amd import ops
    O = _oracle()
    g = torch.Generator().manual_seed(P + 1)
    x = torch.randn(min(P, 256), 8, 519, generator=g)
    if P > 256:
        x = x.repeat(P // 256, 1, 1) + 0.01 * torch.randn(P, 8, 519, generator=g)
    ref = O.mixer(weights_raw, x)
    X = torch.zeros(P * 8, 544)
    X[:, :519] = x.reshape(P * 8, 519)
    out = ops.mixer_fwd(arenas["raw"], X.to(DEV), bf16=True).cpu()
    out32 = ops.mixer_fwd(arenas["raw"], X.to(DEV)).cpu()
    scale = max(1.0, float(ref.abs().max()))
    e_bf16 = float((out - ref).abs().max()) / scale
    e_f32 = float((out32 - ref).abs().max()) / scale
    print(f"P={P}: bf16-operand mixer rel err {e_bf16:.2e} (fp32 path {e_f32:.2e})")
    assert e_f32 < 1e-4
    assert 1e-5 < e_bf16 < 3e-2          # bf16 operand rounding (2^-9 per product) through 25 GEMMs


@pytest.mark.parametrize("S,P", [(1, 5), (3, 64), (5, 33), (12, 40), (16, 7)])
def test_mixer_any_window_length(S, P):
    """pips_mixer_fwd_s: MLPMixer of a Pips(S != 8) (nets/pips.py:93-123 with S tokens) against the oracle -- token MLP S -> 4S -> S,
    head S*130 (zero-padded to a multiple of 4 rows in the arena), exact and split matrix modes."""
    from pips_amd import ops
    from pips_amd.weights import init_state_dict
    O = _oracle()
    sd = init_state_dict(0, S=S, tamed=False)
    arena = ops.pack_weights(sd, torch.device(DEV), S=S)
    g = torch.Generator().manual_seed(100 * S + P)
    x = torch.randn(P, S, 519, generator=g)
    ref = O.mixer(sd, x)
    assert tuple(ref.shape) == (P, S * 130)
    X = torch.zeros(P * S, 544)
    X[:, :519] = x.reshape(P * S, 519)
    scale = max(1.0, float(ref.abs().max()))
    for split in (False, True):
        out = ops.mixer_fwd(arena, X.to(DEV), split=split, S=S).cpu()
        e = float((out - ref).abs().max()) / scale
        print(f"S={S} P={P} split={split}: rel err {e:.2e}")
        assert tuple(out.shape) == (P, S * 130) and e < 1e-4
    lo = ops.mixer_fwd(arena, X.to(DEV), bf16=True, S=S).cpu()
    assert 1e-6 < float((lo - ref).abs().max()) / scale < 3e-2


def test_state_update(weights_raw, arenas):
    from pips_amd import ops
    O = _oracle()
    B, N = 2, 19
    g = torch.Generator().manual_seed(11)
    ffeats = torch.randn(B, 8, N, 128, generator=g)
    coords = torch.rand(B, 8, N, 2, generator=g) * 40
    coords0 = coords + 0.5
    delta = torch.randn(B * N, 8, 130, generator=g)
    ff_ref, co_ref = O.update_step(weights_raw, ffeats, coords, coords0, delta)
    vis_ref = F.linear(ff_ref.reshape(-1, 128), weights_raw["vis_predictor.0.weight"],
                       weights_raw["vis_predictor.0.bias"]).reshape(B, 8, N)
    ff = _pm(ffeats).to(DEV)
    co = _pm(coords).to(DEV)
    traj, vis = ops.state_update(arenas["raw"], delta.reshape(B * N, 1040).to(DEV), ff, co, _pm(coords0).to(DEV),
                                 B, N, 8.0, want_vis=True)
    assert float((ff.cpu() - _pm(ff_ref)).abs().max()) < 2e-5
    assert float((co.cpu() - _pm(co_ref)).abs().max()) < 1e-5
    assert float((traj.cpu() - co_ref * 8.0).abs().max()) < 1e-4
    assert float((vis.cpu() - vis_ref).abs().max()) < 2e-5
    assert torch.equal(traj.cpu()[:, 0], (coords0 * 8.0)[:, 0])                         # frame 0 locked


@pytest.mark.parametrize("src_hw,dst_hw,u8", [((180, 320), (360, 640), True), ((720, 1280), (360, 640), True),
                                              ((360, 640), (360, 640), True), ((97, 131), (368, 496), False)])
def test_resize_frames_matches_interpolate(src_hw, dst_hw, u8):
    """The callers' F.interpolate(rgbs, (H,W), mode='bilinear') (demo.py:26-27) on the device, from decoded uint8
    frames: against ATen on the same values (CPU) to fp32 round-off; identity when the size is unchanged."""
    import os
    import numpy as np
    from pips_amd import ops
    g = torch.Generator().manual_seed(8)
    if src_hw == (180, 320):
        fr = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "demo_half_frames.npz"))["frames"]
        src = torch.from_numpy(fr).permute(0, 3, 1, 2).contiguous().unsqueeze(0)          # (1,8,3,180,320) uint8, real frames
    else:
        src = torch.randint(0, 256, (2, 3, 3) + src_hw, generator=g, dtype=torch.uint8)
    if not u8:
        src = src.float() + torch.rand(src.shape, generator=g)
    ref = F.interpolate(src.float().reshape((-1, 3) + src_hw), dst_hw, mode="bilinear").reshape(src.shape[:-2] + dst_hw)
    got = ops.resize_frames(src.to(DEV), dst_hw).cpu()
    assert got.shape == ref.shape and got.dtype == torch.float32
    err = float((got - ref).abs().max())
    assert err < 1e-4, err
    if src_hw == dst_hw:
        assert torch.equal(got, src.float())


def test_fixed_window_entry_points_equal_the_general_ones(weights_raw):
    """The S = 8 / fixed-mode entry points are thin forms of the general ones: pips_repack_weights(_ex) = pips_repack_weights_s(8),
    pips_encoder_fwd / _bf16 = pips_encoder_fwd_ex(flags), pips_track_ce = pips_track_s(S = 8) -- bit-identical results."""
    import ctypes as C
    from pips_amd import _lib, ops
    from pips_amd.weights import param_table
    lib = _lib.load()
    dev = torch.device(DEV)
    st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
    names = list(param_table().keys())
    srcs = [weights_raw[k].to(dev).contiguous().float() for k in names]
    arr = (C.c_void_p * len(srcs))(*[t.data_ptr() for t in srcs])
    n = lib.pips_weight_arena_bytes() // 4
    a0 = ops.pack_weights(weights_raw, dev)                                  # pips_repack_weights_s(S = 8, all sections)
    a1 = torch.zeros(n, device=dev)
    a2 = torch.zeros(n, device=dev)
    _lib.check(lib.pips_repack_weights(arr, len(srcs), _lib.ptr(a1), st()), "pips_repack_weights")
    _lib.check(lib.pips_repack_weights_ex(arr, len(srcs), _lib.ptr(a2), 7, st()), "pips_repack_weights_ex")
    torch.cuda.synchronize()
    used = a1 != 0                                                           # (alignment gaps are never written)
    assert torch.equal(a1[used], a0[used]) and torch.equal(a2, a1)
    # encoder
    F_, H, W = 8, 128, 160
    rgbs = torch.randint(0, 256, (F_, 3, H, W), generator=torch.Generator().manual_seed(4)).float().to(dev)
    nb = lib.pips_encoder_workspace_bytes(F_, H, W, 8)
    ws = torch.empty(nb // 4, device=dev)
    nlev = lib.pips_pyramid_mirror_offset(F_, H, W, 8)
    for bf16, fn in ((False, lib.pips_encoder_fwd), (True, lib.pips_encoder_fwd_bf16)):
        p1 = torch.zeros(lib.pips_pyramid_floats(F_, H, W, 8), device=dev)
        _lib.check(fn(_lib.ptr(a0), _lib.ptr(rgbs), F_, H, W, 8, _lib.ptr(p1), _lib.ptr(ws), nb, st()), "pips_encoder_fwd")
        p2 = ops.encoder_fwd(a0, rgbs, 8, bf16=bf16)                         # pips_encoder_fwd_ex
        assert torch.equal(p1[:nlev], p2[:nlev]) and float(p1[:nlev].abs().max()) > 0
    # tracker
    B, N = 1, 6
    xys = (torch.rand(B, N, 2, generator=torch.Generator().manual_seed(5)) * 100 + 10).to(dev)
    times = ops.times_table(dev)
    nbt = lib.pips_track_workspace_bytes(B, N)
    assert nbt == lib.pips_track_workspace_bytes_s(B, N, 8)
    outs = []
    for which in (0, 1):
        wst = torch.empty(nbt // 4, device=dev)
        trajs = torch.empty(3, B, 8, N, 2, device=dev); vis = torch.empty(B, 8, N, device=dev); ff = torch.empty(B, N, 128, device=dev)
        args = [_lib.ptr(a0), _lib.ptr(p2), B, 8, H // 8, W // 8, _lib.ptr(xys), None, None, None, _lib.ptr(times), N, 8, 2, 0]
        tail = [_lib.ptr(wst), nbt, _lib.ptr(trajs), _lib.ptr(vis), _lib.ptr(ff), None, None, None, 0, st()]
        rc = lib.pips_track_ce(*args, *tail) if which == 0 else lib.pips_track_s(*args, 8, *tail)
        _lib.check(rc, "pips_track")
        torch.cuda.synchronize()
        outs.append((trajs, vis, ff))
    for x, y in zip(*outs):
        assert torch.equal(x, y)


def test_gather_mfma_batches():
    """gather_mfma_kernel's blocks walk their work items in batches (GM_ENTS = 16 per look-up; BASELINE configs[3] gives a block 8): with
    192 tiles per frame and four frames per XCD a block gets 24+ items, i.e. a second batch whose first item starts from a drained
    pipeline.  A clip's rows depend on nothing but that clip, so the four clips gathered in ONE launch (several batches per block)
    must equal, bit for bit, each clip gathered alone (a single batch per block)."""
    from pips_amd import ops, _lib
    lib = _lib.load()
    B, H8, W8, N = 4, 192, 256, 4096
    F, M = B * 8, B * N * 8
    g = torch.Generator(device=DEV).manual_seed(5)
    per = lib.pips_pyramid_floats(8, H8 * 8, W8 * 8, 8)
    ff = torch.randn(M, 128, generator=g, device=DEV)
    co = torch.rand(M, 2, generator=g, device=DEV) * torch.tensor([W8 + 6.0, H8 + 6.0], device=DEV) - 3.0
    assert lib.pips_gather_route(B, N, H8, W8, 32) == 2 and lib.pips_gather_route(1, N, H8, W8, 32) == 2
    clips, outs = [], []
    for b in range(B):                                   # one clip at a time: 6 items per block
        pyr1 = ops.pyramid_mirror(torch.randn(per, generator=g, device=DEV), 8, H8 * 8, W8 * 8, 8)
        sl = slice(b * N * 8, (b + 1) * N * 8)
        outs.append(ops.mixer_input_build_tiled(pyr1, 1, H8, W8, ff[sl].contiguous(), co[sl].contiguous(), bf16_maps=True).clone())
        clips.append(pyr1)
    # the same maps as ONE four-clip buffer (level-major: every level holds the frames of all clips)
    pyr4 = torch.empty(lib.pips_pyramid_floats(F, H8 * 8, W8 * 8, 8), device=DEV)
    for dst, srcs in zip(ops.pyramid_levels(pyr4, F, H8 * 8, W8 * 8, 8),
                         zip(*[ops.pyramid_levels(p, 8, H8 * 8, W8 * 8, 8) for p in clips])):
        dst.copy_(torch.cat(list(srcs), 0))
    ops.pyramid_mirror(pyr4, F, H8 * 8, W8 * 8, 8)
    X4 = ops.mixer_input_build_tiled(pyr4, B, H8, W8, ff, co, bf16_maps=True)
    X1 = torch.cat(outs, 0)
    assert torch.isfinite(X4).all()
    assert torch.equal(X4, X1)


def test_gather_mfma_repeatable():
    """Two launches of the bf16 matrix-core gather on the same inputs give the same bits (a hardware hazard or a race between the
    product and loader waves shows up as run-to-run differences long before it moves an error norm: round 5's MFMA -> DS read hazard in
    the window scatter did)."""
    from pips_amd import ops, _lib
    lib = _lib.load()
    B, H8, W8, N = 2, 90, 160, 4096
    F, M = B * 8, B * N * 8
    g = torch.Generator(device=DEV).manual_seed(9)
    pyr = ops.pyramid_mirror(torch.randn(lib.pips_pyramid_floats(F, H8 * 8, W8 * 8, 8), generator=g, device=DEV), F, H8 * 8, W8 * 8, 8)
    ff = torch.randn(M, 128, generator=g, device=DEV)
    co = torch.rand(M, 2, generator=g, device=DEV) * torch.tensor([W8 + 6.0, H8 + 6.0], device=DEV) - 3.0
    assert lib.pips_gather_route(B, N, H8, W8, 32) == 2
    X0 = ops.mixer_input_build_tiled(pyr, B, H8, W8, ff, co, bf16_maps=True).clone()
    for _ in range(3):
        X1 = ops.mixer_input_build_tiled(pyr, B, H8, W8, ff, co, bf16_maps=True)
        assert torch.equal(X0, X1)
