"""Thin tensor-level wrappers over the C ABI stage entry points (include/pips_hip.h).

torch is used for device memory and the current stream only; every number is produced by
libpips_hip.so.  These wrappers exist for the parity tests and for callers that keep the
pyramid resident between calls (dense-grid / chained tracking).
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from .weights import param_table

S = 8
LATENT = 128
KIN_PAD = 544
NOUT = 1040


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _f32(t):
    assert t.is_cuda, "pips_amd runs on the GPU only (no CPU fallback)"
    return t.contiguous().to(torch.float32)


PACK_FP32, PACK_BF16, PACK_SPLIT = 1, 2, 4
EPI_BIAS, EPI_GELU, EPI_RESIDUAL, EPI_RES_BF16 = 0, 1, 2, 0x1000          # include/pips_hip.h: PIPS_EPI_*


def pack_more(arena, sections, S=8):
    """Build further sections (PACK_BF16 / PACK_SPLIT) of an arena whose fp32 section is already packed (for window length S)."""
    lib = _lib.load()
    with torch.cuda.device(arena.device):
        _lib.check(lib.pips_repack_weights_s(None, 0, _lib.ptr(arena), int(S), int(sections) & ~PACK_FP32, _stream()),
                   "pips_repack_weights_s")
        # the arena is shared by every stream that drives the module: the new sections must be complete before another
        # thread's stream can read them (pack_weights synchronises for the same reason)
        torch.cuda.current_stream().synchronize()
    return arena


def pack_weights(state_dict, device, sections=PACK_FP32 | PACK_BF16 | PACK_SPLIT, S=8) -> torch.Tensor:
    """state dict (reference key names/layouts, of a ``Pips(S=S)``) -> packed device arena (pips_repack_weights_s);
    ``sections``: which of the fp32 / bf16-copy / split-plane sections to build now (pack_more adds the others later)."""
    lib = _lib.load()
    table = param_table(S)
    names = list(table.keys())
    missing = [k for k in names if k not in state_dict]
    if missing:
        raise KeyError(f"state dict lacks {len(missing)} tensors, e.g. {missing[:3]}")
    for k in names:           # the C side sees bare pointers: a checkpoint of another window length must not get that far
        if tuple(state_dict[k].shape) != tuple(table[k][0]):
            raise ValueError(f"{k}: shape {tuple(state_dict[k].shape)}, a Pips(S={S}) holds {tuple(table[k][0])}")
    nbytes = lib.pips_weight_arena_bytes_s(int(S))
    if nbytes == 0:
        raise ValueError(f"window length S={S} is outside 1..32")
    with torch.cuda.device(device):
        srcs = [_f32(state_dict[k].detach().to(device)) for k in names]
        arena = torch.empty(nbytes // 4, dtype=torch.float32, device=device)
        arr = (C.c_void_p * len(srcs))(*[t.data_ptr() for t in srcs])
        _lib.check(lib.pips_repack_weights_s(arr, len(srcs), _lib.ptr(arena), int(S), int(sections) | PACK_FP32, _stream()),
                   "pips_repack_weights_s")
        torch.cuda.current_stream().synchronize()      # srcs may be temporaries
    return arena


def times_table(device, S=8) -> torch.Tensor:
    # torch.linspace(0, S, S) exactly as nets/pips.py:519 builds it
    return torch.linspace(0, S, S, device=device, dtype=torch.float32)


def pyramid_levels(pyr: torch.Tensor, F: int, H: int, W: int, stride: int):
    """Views (F,H_l,W_l,128) of the packed pyramid buffer."""
    lib = _lib.load()
    out = []
    h, w = H // stride, W // stride
    for l in range(4):
        off = lib.pips_pyramid_offset(F, H, W, stride, l)
        out.append(pyr[off:off + F * h * w * LATENT].view(F, h, w, LATENT))
        h, w = h // 2, w // 2
    return out


def encoder_fwd(arena, rgbs, stride, bf16=False, split=False):
    """rgbs (F,3,H,W) 0..255 -> packed channel-last pyramid buffer.  bf16: bf16 conv operands;
    split: fp32-grade split-bf16 convolutions."""
    lib = _lib.load()
    u8 = rgbs.dtype == torch.uint8                    # decoded frames go in as they are
    rgbs = rgbs.contiguous() if u8 else _f32(rgbs)
    F, _, H, W = rgbs.shape
    with torch.cuda.device(rgbs.device):
        pyr = torch.empty(lib.pips_pyramid_floats(F, H, W, stride), dtype=torch.float32, device=rgbs.device)
        nb = lib.pips_encoder_workspace_bytes(F, H, W, stride)
        ws = torch.empty(nb // 4, dtype=torch.float32, device=rgbs.device)
        flags = (4 if bf16 else 0) | (8 if u8 else 0) | (16 if split else 0)   # PIPS_FLAG_BF16_ENCODER | RGB_U8 | SPLIT_BF16
        _lib.check(lib.pips_encoder_fwd_ex(_lib.ptr(arena), _lib.ptr(rgbs), F, H, W, stride, flags, _lib.ptr(pyr),
                                           _lib.ptr(ws), nb, _stream()), "pips_encoder_fwd_ex")
    return pyr


def resize_frames(rgbs, size):
    """uint8 or float frames (..., 3, h, w) -> float32 (..., 3, H, W), values 0..255: the on-device form of the callers'
    ``F.interpolate(rgbs, (H, W), mode='bilinear')`` (demo.py:26-27).  Feed the result to ``Pips.forward``."""
    lib = _lib.load()
    H, W = int(size[0]), int(size[1])
    h, w = rgbs.shape[-2:]
    src = rgbs.contiguous() if rgbs.dtype == torch.uint8 else _f32(rgbs)
    planes = src.numel() // (h * w)
    out = torch.empty(tuple(rgbs.shape[:-2]) + (H, W), dtype=torch.float32, device=rgbs.device)
    with torch.cuda.device(rgbs.device):
        _lib.check(lib.pips_resize_frames(_lib.ptr(src), 1 if src.dtype == torch.uint8 else 0, planes, h, w, _lib.ptr(out),
                                          H, W, _stream()), "pips_resize_frames")
    return out


def point_sample(level0, B, xy):
    """level0 (B*S,H8,W8,128), xy (B,N,2) map pixels -> (B,N,128)."""
    lib = _lib.load()
    xy = _f32(xy)
    F, H8, W8, _ = level0.shape
    N = xy.shape[1]
    out = torch.empty(B, N, LATENT, dtype=torch.float32, device=xy.device)
    with torch.cuda.device(xy.device):
        _lib.check(lib.pips_point_sample(_lib.ptr(level0), B, F // B, H8, W8, _lib.ptr(xy), N, _lib.ptr(out),
                                         _stream()), "pips_point_sample")
    return out


def mixer_input_build(pyr, B, H8, W8, ffeats, coords, bf16_maps=False):
    """ffeats (B*N*S,128), coords (B*N*S,2) particle-major -> X (B*N*S, 544).  bf16_maps: the gather reads the bf16 mirror
    behind the fp32 levels of ``pyr`` (PIPS_FLAG_BF16_MAPS; pyramid_mirror() writes it)."""
    lib = _lib.load()
    if bf16_maps:
        ffeats, coords = _f32(ffeats), _f32(coords)
        M = ffeats.shape[0]
        N = M // (B * S)
        X = torch.empty(M, KIN_PAD, dtype=torch.float32, device=ffeats.device)
        tt = times_table(ffeats.device)
        with torch.cuda.device(ffeats.device):
            _lib.check(lib.pips_mixer_input_build_ex(_lib.ptr(pyr), B, S, H8, W8, _lib.ptr(ffeats), _lib.ptr(coords), _lib.ptr(tt),
                                                     N, None, 32, _lib.ptr(X), _stream()), "pips_mixer_input_build_ex")
        return X
    ffeats, coords = _f32(ffeats), _f32(coords)
    M = ffeats.shape[0]
    N = M // (B * S)
    X = torch.empty(M, KIN_PAD, dtype=torch.float32, device=ffeats.device)
    tt = times_table(ffeats.device)
    with torch.cuda.device(ffeats.device):
        _lib.check(lib.pips_mixer_input_build(_lib.ptr(pyr), B, S, H8, W8, _lib.ptr(ffeats), _lib.ptr(coords),
                                              _lib.ptr(tt), N, _lib.ptr(X), _stream()), "pips_mixer_input_build")
    return X


def pyramid_mirror(pyr, F, H, W, stride):
    """(re)write the bf16 mirror of a packed pyramid buffer from its fp32 levels (pips_pyramid_mirror)"""
    lib = _lib.load()
    with torch.cuda.device(pyr.device):
        _lib.check(lib.pips_pyramid_mirror(_lib.ptr(pyr), F, H, W, stride, _stream()), "pips_pyramid_mirror")
    return pyr


def mixer_input_build_tiled(pyr, B, H8, W8, ffeats, coords, out=None, bf16_maps=False):
    """Same as mixer_input_build through the tiled kernels for dense query sets.  bf16_maps: the bf16 mode's matrix-core kernel on
    the bf16 mirror behind the fp32 levels of ``pyr`` (PIPS_FLAG_BF16_MAPS; features rounded to bf16 as well, like the
    reference under autocast)."""
    lib = _lib.load()
    ffeats, coords = _f32(ffeats), _f32(coords)
    M = ffeats.shape[0]
    N = M // (B * S)
    X = out if out is not None else torch.empty(M, KIN_PAD, dtype=torch.float32, device=ffeats.device)
    tt = times_table(ffeats.device)
    nb = lib.pips_gather_scratch_bytes(B, N, H8, W8)
    scratch = torch.empty(nb, dtype=torch.uint8, device=ffeats.device)
    with torch.cuda.device(ffeats.device):
        _lib.check(lib.pips_mixer_input_build_tiled_ex(_lib.ptr(pyr), B, S, H8, W8, _lib.ptr(ffeats), _lib.ptr(coords),
                                                       _lib.ptr(tt), N, 32 if bf16_maps else 0, _lib.ptr(X), _lib.ptr(scratch), nb,
                                                       _stream(), None), "pips_mixer_input_build_tiled_ex")
    return X


def mixer_input_build_tiled_timed(pyr, B, H8, W8, ffeats, coords, bf16_maps=False):
    """(X, {"bin": ms, "embed": ms, "gather": ms}): HIP-event durations of the three launches of the tiled path."""
    lib = _lib.load()
    ffeats, coords = _f32(ffeats), _f32(coords)
    M = ffeats.shape[0]
    N = M // (B * S)
    X = torch.empty(M, KIN_PAD, dtype=torch.float32, device=ffeats.device)
    tt = times_table(ffeats.device)
    nb = lib.pips_gather_scratch_bytes(B, N, H8, W8)
    scratch = torch.empty(nb, dtype=torch.uint8, device=ffeats.device)
    ms = (C.c_float * 3)()
    with torch.cuda.device(ffeats.device):
        _lib.check(lib.pips_mixer_input_build_tiled_ex(_lib.ptr(pyr), B, S, H8, W8, _lib.ptr(ffeats), _lib.ptr(coords),
                                                       _lib.ptr(tt), N, 32 if bf16_maps else 0, _lib.ptr(X), _lib.ptr(scratch), nb,
                                                       _stream(), ms), "pips_mixer_input_build_tiled_ex")
    return X, {"bin": ms[0], "embed": ms[1], "gather": ms[2]}


def score_map_terms(pyr, B, H8, W8, ffeats, tgt):
    """Per mixer row {loss at the target pixel, sum of the losses of the other pixels} of the dense score map
    (nets/pips.py:501-511 + score_map_loss :58-92).  ffeats (B*N*S,128), tgt (B*N*S,3) = {x, y, use}."""
    lib = _lib.load()
    ffeats, tgt = _f32(ffeats), _f32(tgt)
    M = ffeats.shape[0]
    N = M // (B * S)
    U = torch.empty(lib.pips_score_map_workspace_bytes(B, S, H8, W8) // 4, dtype=torch.float32, device=ffeats.device)
    out = torch.empty(M, 2, dtype=torch.float32, device=ffeats.device)
    with torch.cuda.device(ffeats.device):
        _lib.check(lib.pips_score_map_prepare(_lib.ptr(pyr), B, S, H8, W8, _lib.ptr(U), _stream()), "pips_score_map_prepare")
        _lib.check(lib.pips_score_map_terms(_lib.ptr(U), B, S, H8, W8, _lib.ptr(ffeats), N, _lib.ptr(tgt), _lib.ptr(out),
                                            _stream()), "pips_score_map_terms")
    return out


def mixer_fwd(arena, X, bf16=False, split=False, S=8, stream_bf16=False):
    """X (M,544) -> delta (M/S, S*130).  bf16: bf16 MFMA operands in the channel-mix/head GEMMs;
    split: every GEMM on the fp32-grade split-bf16 path.  S != 8 (arena packed for that S):
    pips_mixer_fwd_s, whose rows are pips_delta_stride(S) apart (cut back to S*130 here).
    stream_bf16 (with bf16, S = 8): the residual stream is a bf16 tensor (PIPS_FLAG_BF16_STREAM)."""
    lib = _lib.load()
    X = _f32(X)
    M = X.shape[0]
    if stream_bf16:
        assert bf16 and S == 8 and not split
        delta = torch.empty(M // S, NOUT, dtype=torch.float32, device=X.device)
        nb = lib.pips_mixer_workspace_bytes(M)
        ws = torch.empty(nb // 4, dtype=torch.float32, device=X.device)
        with torch.cuda.device(X.device):
            _lib.check(lib.pips_mixer_fwd_s(_lib.ptr(arena), _lib.ptr(X), M, 8, 2 | 64, _lib.ptr(delta), _lib.ptr(ws), nb, _stream()),
                       "pips_mixer_fwd_s")
        return delta
    if S != 8:
        ld = lib.pips_delta_stride(int(S))
        delta = torch.empty(M // S, ld, dtype=torch.float32, device=X.device)
        nb = lib.pips_mixer_workspace_bytes_s(M, int(S))
        ws = torch.empty(nb // 4, dtype=torch.float32, device=X.device)
        with torch.cuda.device(X.device):
            _lib.check(lib.pips_mixer_fwd_s(_lib.ptr(arena), _lib.ptr(X), M, int(S), 16 if split else (2 if bf16 else 0),
                                            _lib.ptr(delta), _lib.ptr(ws), nb, _stream()), "pips_mixer_fwd_s")
        return delta[:, :S * 130]
    delta = torch.empty(M // S, NOUT, dtype=torch.float32, device=X.device)
    nb = lib.pips_mixer_workspace_bytes(M)
    ws = torch.empty(nb // 4, dtype=torch.float32, device=X.device)
    with torch.cuda.device(X.device):
        fn = lib.pips_mixer_fwd_x3 if split else (lib.pips_mixer_fwd_bf16 if bf16 else lib.pips_mixer_fwd)
        _lib.check(fn(_lib.ptr(arena), _lib.ptr(X), M, _lib.ptr(delta), _lib.ptr(ws), nb, _stream()), "pips_mixer_fwd")
    return delta


def mixer_fwd_timed(arena, X, flags=0):
    """Profiling: one mixer pass with HIP events around every GEMM launch.
    flags: 0 exact fp32, 2 bf16 operands, 16 split-bf16 (PIPS_FLAG_*).
    Returns (delta, {in_proj, up_proj, down_proj, head} milliseconds per launch)."""
    lib = _lib.load()
    X = _f32(X)
    M = X.shape[0]
    delta = torch.empty(M // S, NOUT, dtype=torch.float32, device=X.device)
    nb = lib.pips_mixer_workspace_bytes(M)
    ws = torch.empty(nb // 4, dtype=torch.float32, device=X.device)
    ms = (C.c_float * 5)()
    with torch.cuda.device(X.device):
        _lib.check(lib.pips_mixer_fwd_timed_ex(_lib.ptr(arena), _lib.ptr(X), M, flags, _lib.ptr(delta), _lib.ptr(ws), nb,
                                               _stream(), ms), "pips_mixer_fwd_timed_ex")
    return delta, {"in_proj": ms[0], "up_proj": ms[1], "down_proj": ms[2], "head": ms[3], "event_overhead": ms[4]}


def mixer_gemm_train(arena, X, flags=0, reps=4):
    """Profiling: a mixer pass on X, then the 12 up-projections / 12 down-projections of the pass as back-to-back launch
    trains between ONE event pair each (pips_mixer_gemm_train).  Returns {up_proj, down_proj} milliseconds per launch,
    start to start -- durations that tile the forward's timeline (no per-launch markers, nothing subtracted)."""
    lib = _lib.load()
    X = _f32(X)
    M = X.shape[0]
    delta = torch.empty(M // S, NOUT, dtype=torch.float32, device=X.device)
    nb = lib.pips_mixer_workspace_bytes(M)
    ws = torch.empty(nb // 4, dtype=torch.float32, device=X.device)
    ms = (C.c_float * 2)()
    with torch.cuda.device(X.device):
        _lib.check(lib.pips_mixer_fwd_s(_lib.ptr(arena), _lib.ptr(X), M, S, flags, _lib.ptr(delta), _lib.ptr(ws), nb, _stream()),
                   "pips_mixer_fwd_s")
        _lib.check(lib.pips_mixer_gemm_train(_lib.ptr(arena), M, flags, _lib.ptr(ws), nb, _stream(), reps, ms),
                   "pips_mixer_gemm_train")
    return {"up_proj": ms[0], "down_proj": ms[1]}


def state_update(arena, delta, ffeats, coords, coords0, B, N, stride, want_vis=False):
    """In-place update of ffeats/coords (particle-major); returns (traj (B,S,N,2) px, vis or None)."""
    lib = _lib.load()
    traj = torch.empty(B, S, N, 2, dtype=torch.float32, device=delta.device)
    vis = torch.empty(B, S, N, dtype=torch.float32, device=delta.device) if want_vis else None
    with torch.cuda.device(delta.device):
        _lib.check(lib.pips_state_update(_lib.ptr(arena), _lib.ptr(delta), _lib.ptr(ffeats), _lib.ptr(coords),
                                         _lib.ptr(coords0), B, N, float(stride), _lib.ptr(traj), _lib.ptr(vis),
                                         _stream()), "pips_state_update")
    return traj, vis


def gemm(A, W, bias=None, epi=0, R=None):
    """C = epi(A @ W.T + bias).  epi: 0 none, 1 GELU, 2 + R."""
    lib = _lib.load()
    A, W = _f32(A), _f32(W)
    M, K = A.shape
    N = W.shape[0]
    Cm = torch.empty(M, N, dtype=torch.float32, device=A.device)
    with torch.cuda.device(A.device):
        _lib.check(lib.pips_gemm_f32(_lib.ptr(A), K, _lib.ptr(W), _lib.ptr(bias), _lib.ptr(Cm), N, M, N, K, epi,
                                     _lib.ptr(R), N if R is not None else 0, _stream()), "pips_gemm_f32")
    return Cm


def split_bf16x3(w):
    """fp32 tensor -> its three bf16 planes, int16 tensor of shape (3, *w.shape) (exact split: each plane is the round-to-nearest-even bf16 of the
    remainder, so the three sum to w and the dropped cross terms are zero-mean; gemm_x3.hip)."""
    lib = _lib.load()
    w = _f32(w)
    out = torch.empty((3,) + tuple(w.shape), dtype=torch.int16, device=w.device)
    with torch.cuda.device(w.device):
        _lib.check(lib.pips_split_bf16x3(_lib.ptr(w), w.numel(), _lib.ptr(out), _stream()), "pips_split_bf16x3")
    return out


def gemm_x3(A, W3, bias=None, epi=0, R=None):
    """gemm() on the split-bf16 path; W3 = split_bf16x3(W) with W of shape (N, K)."""
    lib = _lib.load()
    A = _f32(A)
    M, K = A.shape
    N = W3.shape[1]
    Cm = torch.empty(M, N, dtype=torch.float32, device=A.device)
    with torch.cuda.device(A.device):
        _lib.check(lib.pips_gemm_f32x3(_lib.ptr(A), K, _lib.ptr(W3), _lib.ptr(bias), _lib.ptr(Cm), N, M, N, K, epi,
                                       _lib.ptr(R), N if R is not None else 0, _stream()), "pips_gemm_f32x3")
    return Cm


def gemm_bf16(A, W, bias=None, epi=0, R=None, out_bf16=False):
    """gemm() with bf16 MFMA operands: A fp32 or bfloat16 (M,K), W bfloat16 (N,K); returns fp32 or bfloat16 (M,N)."""
    lib = _lib.load()
    assert W.dtype == torch.bfloat16 and A.dtype in (torch.float32, torch.bfloat16)
    A, W = A.contiguous(), W.contiguous()
    M, K = A.shape
    N = W.shape[0]
    Cm = torch.empty(M, N, dtype=torch.bfloat16 if out_bf16 else torch.float32, device=A.device)
    if R is not None and R.dtype == torch.bfloat16:      # a bf16 residual (with a bf16 output): the mixer's bf16 residual stream
        assert out_bf16 and epi == 2
        epi = epi | EPI_RES_BF16
        R = R.contiguous()
    with torch.cuda.device(A.device):
        _lib.check(lib.pips_gemm_bf16(_lib.ptr(A), int(A.dtype == torch.bfloat16), K, _lib.ptr(W), _lib.ptr(bias), _lib.ptr(Cm),
                                      int(out_bf16), N, M, N, K, epi, _lib.ptr(R), N if R is not None else 0, _stream()),
                   "pips_gemm_bf16")
    return Cm


def partial_sums(stats):
    """Pivoted InstanceNorm partials (F, parts, C, 4) = {sum(x-p), sum((x-p)^2), p, n} -> fp64 (sum x, sum x^2) per (F, C)."""
    st = stats.double()
    s, q, p, n = st[..., 0], st[..., 1], st[..., 2], st[..., 3]
    p = torch.where(n > 0, p, torch.zeros_like(p))
    return (n * p + s).sum(dim=1), (q + 2 * p * s + n * p * p).sum(dim=1)


def conv_nhwc(x, w_packed, bias, ksize, stride, pad, want_stats=False):
    """x (F,H,W,Cin) NHWC, w_packed (Cout, k, k, Cin) -> (F,Ho,Wo,Cout) [+ pivoted partial stats (F, parts, Cout, 4),
    parts = m tiles x wave rows: see partial_sums()]."""
    lib = _lib.load()
    x, w_packed = _f32(x), _f32(w_packed)
    F, H, W, Cin = x.shape
    Cout = w_packed.shape[0]
    Ho = (H + 2 * pad - ksize) // stride + 1
    Wo = (W + 2 * pad - ksize) // stride + 1
    out = torch.empty(F, Ho, Wo, Cout, dtype=torch.float32, device=x.device)
    stats = None
    if want_stats:
        stats = torch.zeros(F, 2 * ((Ho * Wo + 63) // 64) + 4, Cout, 4, dtype=torch.float32, device=x.device)
    tiles = C.c_int(0)
    with torch.cuda.device(x.device):
        _lib.check(lib.pips_conv_nhwc_f32(_lib.ptr(x), F, H, W, Cin, _lib.ptr(w_packed), _lib.ptr(bias), Cout, ksize,
                                          stride, pad, _lib.ptr(out), _lib.ptr(stats), C.byref(tiles), _stream()),
                   "pips_conv_nhwc_f32")
    if want_stats:
        return out, stats.view(-1)[: F * tiles.value * Cout * 4].view(F, tiles.value, Cout, 4)
    return out


def conv_nhwc_bf16(x, w_bf16, bias, ksize, stride, pad, want_stats=False):
    """conv_nhwc() with bf16 MFMA operands; w_bf16 = w_packed.bfloat16() of shape (Cout, k, k, Cin)."""
    lib = _lib.load()
    x = _f32(x)
    assert w_bf16.dtype == torch.bfloat16
    w_bf16 = w_bf16.contiguous()
    F, H, W, Cin = x.shape
    Cout = w_bf16.shape[0]
    Ho = (H + 2 * pad - ksize) // stride + 1
    Wo = (W + 2 * pad - ksize) // stride + 1
    out = torch.empty(F, Ho, Wo, Cout, dtype=torch.float32, device=x.device)
    stats = None
    if want_stats:
        stats = torch.zeros(F, 2 * ((Ho * Wo + 63) // 64) + 4, Cout, 4, dtype=torch.float32, device=x.device)
    tiles = C.c_int(0)
    with torch.cuda.device(x.device):
        _lib.check(lib.pips_conv_nhwc_bf16(_lib.ptr(x), F, H, W, Cin, _lib.ptr(w_bf16), _lib.ptr(bias), Cout, ksize, stride,
                                           pad, _lib.ptr(out), _lib.ptr(stats), C.byref(tiles), _stream()),
                   "pips_conv_nhwc_bf16")
    if want_stats:
        return out, stats.view(-1)[: F * tiles.value * Cout * 4].view(F, tiles.value, Cout, 4)
    return out


def conv_nhwc_bf16_maps(x_bf16, w_bf16, bias, ksize, stride, pad, in_norm=None, out_bf16=True, want_stats=False):
    """The same convolution on a bf16 NHWC map (the bf16 encoder's form): ``in_norm`` (F, Cin, 2) = {mean, rstd} of the
    producing layer applies relu((x - mean) * rstd) while the map is staged (64 -> 64 3x3 layers the LDS-resident kernel
    takes); the output map is bf16 or fp32."""
    lib = _lib.load()
    assert x_bf16.dtype == torch.bfloat16 and w_bf16.dtype == torch.bfloat16 and x_bf16.is_cuda
    x_bf16, w_bf16 = x_bf16.contiguous(), w_bf16.contiguous()
    F, H, W, Cin = x_bf16.shape
    Cout = w_bf16.shape[0]
    Ho = (H + 2 * pad - ksize) // stride + 1
    Wo = (W + 2 * pad - ksize) // stride + 1
    out = torch.empty(F, Ho, Wo, Cout, dtype=torch.bfloat16 if out_bf16 else torch.float32, device=x_bf16.device)
    nrm = None if in_norm is None else _f32(in_norm)
    stats = None
    cap = max(2 * ((Ho * Wo + 63) // 64) + 4, ((Wo + 31) // 32) * ((Ho + 3) // 4) * 4)
    if want_stats:
        stats = torch.zeros(F, cap, Cout, 4, dtype=torch.float32, device=x_bf16.device)
    tiles = C.c_int(0)
    with torch.cuda.device(x_bf16.device):
        _lib.check(lib.pips_conv_nhwc_bf16_maps(_lib.ptr(x_bf16), _lib.ptr(nrm), F, H, W, Cin, _lib.ptr(w_bf16), _lib.ptr(bias),
                                                Cout, ksize, stride, pad, _lib.ptr(out), 1 if out_bf16 else 0, _lib.ptr(stats),
                                                cap, C.byref(tiles), _stream()), "pips_conv_nhwc_bf16_maps")
    if want_stats:
        return out, stats.view(-1)[: F * tiles.value * Cout * 4].view(F, tiles.value, Cout, 4)
    return out


def conv_nhwc_x3(x, w3, bias, ksize, stride, pad, want_stats=False):
    """conv_nhwc() on the split-bf16 path; w3 = split_bf16x3(w_packed)."""
    lib = _lib.load()
    x = _f32(x)
    F, H, W, Cin = x.shape
    Cout = w3.shape[1]
    Ho = (H + 2 * pad - ksize) // stride + 1
    Wo = (W + 2 * pad - ksize) // stride + 1
    out = torch.empty(F, Ho, Wo, Cout, dtype=torch.float32, device=x.device)
    stats = None
    if want_stats:
        stats = torch.zeros(F, 2 * ((Ho * Wo + 63) // 64) + 4, Cout, 4, dtype=torch.float32, device=x.device)
    tiles = C.c_int(0)
    with torch.cuda.device(x.device):
        _lib.check(lib.pips_conv_nhwc_f32x3(_lib.ptr(x), F, H, W, Cin, _lib.ptr(w3), _lib.ptr(bias), Cout, ksize,
                                            stride, pad, _lib.ptr(out), _lib.ptr(stats), C.byref(tiles), _stream()),
                   "pips_conv_nhwc_f32x3")
    if want_stats:
        return out, stats.view(-1)[: F * tiles.value * Cout * 4].view(F, tiles.value, Cout, 4)
    return out
