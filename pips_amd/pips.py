"""``Pips`` -- drop-in for ``nets.pips.Pips`` (nets/pips.py:400-611) whose forward runs
entirely in libpips_hip.so (hand-written HIP for gfx950).

Same constructor ``Pips(S=8, stride=8)``, same 200-key state dict (so
``saverloader.load`` / ``load_state_dict`` of a reference checkpoint work unchanged),
same ``forward(xys, rgbs, coords_init, feat_init, iters, trajs_g, vis_g, valids, sw,
return_feat, is_train)`` signature and the same returned tuple:
``(coord_predictions [iters x (B,S,N,2)], coord_predictions2 [iters+4], vis_e (B,S,N),
losses)`` or, with ``return_feat=True``, ``(..., vis_e, ffeat (B,N,128), losses)``.

Inference only: ``is_train=True`` raises.  A summary writer ``sw`` is accepted and never drawn into: the reference uses it
only to DRAW (nets/pips.py:447,477-497,541-557,564-598 -- none of those branches feeds the returned tuple), so on the samples
where the callers' writer has ``save_this`` set (every ``log_freq``-th, test_on_flt.py:197,267-272) the forward warns once
and returns exactly what it returns for ``sw=None``; ``losses`` carries the reference's ``(seq_loss, vis_loss, ce_loss)`` when ``trajs_g`` is given
(nets/pips.py:600-606; the score-map loss is reduced on the fly by ``pips_forward_ce``).  ``S`` = 8 (the window of every
shipped checkpoint and caller) runs kernels specialised for it; any other ``1 <= S <= 32`` runs the token mixing, the
final LayerNorm and the state update on generic HIP kernels (``pips_*_s`` entry points).  There is no
PyTorch fallback: without the HIP library or a GPU the forward raises.

Weights are repacked for the kernels when a parameter's storage or version counter changes
(``load_state_dict``, ``.to()``, in-place ops).  Writes through ``param.data`` bump neither: call
``invalidate_weights()`` after such surgery.
"""
from __future__ import annotations

import ctypes as C
import threading

import torch
import torch.nn as nn

from . import _lib, ops
from .weights import init_state_dict, param_table


class FeatureCache:
    """Encoder output kept on the device for re-use: the packed 4-level channel-last pyramid of
    ``B`` clips x ``T`` frames (``T`` need not be 8).  Produced by ``Pips.encode``, consumed by
    ``Pips.track``.  Per-frame InstanceNorm (nets/pips.py:153-157) makes a frame's maps
    independent of which clip/window it sits in, so the cache is exact for any window."""

    def __init__(self, pyr, B, T, H, W, stride, bf16_maps=False):
        self.pyr, self.B, self.T, self.H, self.W, self.stride = pyr, B, T, H, W, stride
        self.bf16_maps = bf16_maps          # the buffer's bf16 mirror is valid (written by the bf16 encoder): PIPS_FLAG_BF16_MAPS

    @property
    def map_size(self):
        return self.H // self.stride, self.W // self.stride


class _Node(nn.Module):
    """Bare container: only carries parameters/children under the reference's names."""


class Pips(nn.Module):
    def __init__(self, S: int = 8, stride: int = 8):
        super().__init__()
        if not 1 <= int(S) <= 32:
            # the reference builds S-dependent mixer weights for any S (nets/pips.py:295-301); here S = 8 (every shipped
            # checkpoint / caller) runs specialised kernels and 1..32 (PIPS_S_MAX) generic ones
            raise ValueError("pips_amd.Pips supports window lengths S = 1..32")
        self.S = S = int(S)
        self.stride = stride
        self.hidden_dim = 256
        self.latent_dim = 128
        self.corr_levels = 4
        self.corr_radius = 3
        init = init_state_dict(seed=0, S=S)
        for name in param_table(S):
            *path, leaf = name.split(".")
            node = self
            for p in path:
                if p not in node._modules:
                    node.add_module(p, _Node())
                node = node._modules[p]
            node.register_parameter(leaf, nn.Parameter(init[name], requires_grad=False))
        # torch.float32 (default) or torch.bfloat16: operand type of the mixer's channel-mix / head
        # GEMMs (BASELINE config 3).  Also switched on by an enclosing
        # ``torch.autocast("cuda", dtype=torch.bfloat16)``, the way the reference would be run in bf16.
        self.mixer_dtype = torch.float32
        self.encoder_dtype = torch.float32          # same switch for the encoder's 3x3 / 1x1 convolutions
        # residual stream of the bf16 mixer (S = 8).  None (default): follows the mixer -- a bf16 mixer holds a bf16 stream, as
        # PreNormResidual's `fn(norm(x)) + x` does under autocast (nets/pips.py:93-100; PIPS_FLAG_BF16_STREAM; round 5: 1.4e-2 px
        # against the reference arithmetic under autocast at BASELINE configs[2], the fp32 stream 1.5e-2).  torch.float32 keeps it fp32 (rounds 1-4).
        self.mixer_stream_dtype = None
        # "exact": fp32 MFMA (products and sums bitwise an fmaf chain).  "split": the fp32-grade
        # split-bf16 matrix path (PIPS_FLAG_SPLIT_BF16: three exact bf16 terms per fp32 operand, six
        # bf16 products per fp32 product, fp32 accumulation) -- same accuracy class, ~1.2x faster.
        self.matmul = "exact"
        self._names = list(param_table(S).keys())
        self._plist = None
        self._warned_precedence = False
        self._warned_sw = False
        self._arena = None
        self._arena_key = None
        self._arena_params = None
        self._arena_sections = 0
        self._ws = {}
        self._times = None
        # The reference module is stateless between calls; this one caches the packed weights and its scratch memory.
        # Both are guarded: the lock covers (re)packing and the workspace tables, and scratch is kept PER STREAM, so
        # Python threads driving one module on different streams never share a workspace (kernels of two streams run
        # concurrently on the device; a lock around the launch would not keep them apart).
        self._lock = threading.RLock()

    # ------------------------------------------------------------------ weights
    def invalidate_weights(self):
        """Drop the packed kernel-side copy of the weights; the next forward repacks.  Needed only after
        writes that bypass autograd's version counter (``p.data.copy_()``, ``p.data.mul_()``)."""
        self._arena = self._arena_key = self._plist = self._arena_params = None

    def _apply(self, fn, *a, **kw):                     # .to() / .cuda() / .float(): parameters may be re-created
        out = super()._apply(fn, *a, **kw)
        self.invalidate_weights()
        return out

    def _packed(self, device, need=None):
        """The packed weight arena for ``device``.  Only the sections the current matrix mode reads are built (fp32 always;
        bf16 copies for the bf16-operand modes; split planes for matmul='split'): a rank of a bf16 job never packs the 171 MB
        of split planes.  ``need`` = ops.PACK_* mask (default: what self._flags() implies)."""
        if need is None:
            fl = self._flags()
            need = ops.PACK_FP32 | (ops.PACK_BF16 if fl & 6 else 0) | (ops.PACK_SPLIT if fl & 16 else 0)
        if self._plist is None:
            # (owning module, leaf name) of every parameter: the LIVE object is looked up on every forward, so a
            # parameter that was replaced (load_state_dict(assign=True), ``node.weight = nn.Parameter(...)``) is seen
            # like one that was mutated in place
            self._plist = []
            for name in self._names:
                *path, leaf = name.split(".")
                node = self
                for q in path:
                    node = node._modules[q]
                self._plist.append((node._parameters, leaf))
        live = [d[leaf] for d, leaf in self._plist]
        key = (str(device),) + tuple((id(p), p.data_ptr(), p._version) for p in live)
        if self._arena is None or key != self._arena_key:
            self._arena = ops.pack_weights(dict(zip(self._names, live)), device, sections=need, S=self.S)
            self._arena_key = key
            self._arena_sections = need | ops.PACK_FP32
            self._arena_params = live          # keeps the ids in the key from being recycled
        elif need & ~self._arena_sections:
            ops.pack_more(self._arena, need & ~self._arena_sections, S=self.S)
            self._arena_sections |= need
        return self._arena

    def _flags(self):
        ac = torch.is_autocast_enabled("cuda") and torch.get_autocast_dtype("cuda") == torch.bfloat16
        if self.matmul not in ("exact", "split"):
            raise ValueError(f"Pips.matmul must be 'exact' or 'split', not {self.matmul!r}")
        bf = (2 if ac or self.mixer_dtype == torch.bfloat16 else 0) | \
             (4 if ac or self.encoder_dtype == torch.bfloat16 else 0)     # PIPS_FLAG_BF16_MIXER | _ENCODER
        if (bf & 2) and self.S == 8 and self.mixer_stream_dtype in (None, torch.bfloat16):
            bf |= 64                                                      # PIPS_FLAG_BF16_STREAM
        if bf and self.matmul == "split" and not self._warned_precedence:
            # a bf16 request (autocast or mixer_dtype / encoder_dtype) wins over matmul="split": say so once
            import warnings
            warnings.warn("pips_amd.Pips: bf16 operands requested (autocast or *_dtype = bfloat16): matmul='split' is "
                          "ignored for those stages", stacklevel=3)
            self._warned_precedence = True
        return bf if bf or self.matmul == "exact" else 16                 # PIPS_FLAG_SPLIT_BF16

    def _workspace(self, lib, dims, device):
        """Scratch of one forward: one buffer per (device, stream), replaced when the problem size changes."""
        slot = ("fwd", str(device), int(torch.cuda.current_stream(device).cuda_stream))
        with self._lock:
            ent = self._ws.get(slot)
            if ent is None or ent[0] != dims:
                nb = lib.pips_workspace_bytes(*dims)
                if nb == 0:
                    raise _lib.PipsHipError(f"unsupported problem size {dims}")
                self._ws.pop(slot, None)
                ent = self._ws[slot] = (dims, torch.empty(nb // 4, dtype=torch.float32, device=device))
            return ent[1]

    def _aux(self, dev):
        """(packed weights, frame-time table) for ``dev`` -- built or refreshed under the module lock."""
        with self._lock:
            arena = self._packed(dev)
            if self._times is None or self._times.device != dev:
                self._times = ops.times_table(dev, self.S)
                # shared by every stream that drives the module: complete before another thread's stream reads it
                torch.cuda.current_stream(dev).synchronize()
            return arena, self._times

    # ------------------------------------------------------------------ forward
    @torch.no_grad()
    def forward(self, xys, rgbs, coords_init=None, feat_init=None, iters=3, trajs_g=None, vis_g=None,
                valids=None, sw=None, return_feat=False, is_train=False):
        if is_train:
            raise NotImplementedError("pips_amd.Pips is the inference path (nets/pips.py:535: is_train=False)")
        if sw is not None and getattr(sw, "save_this", False) and not self._warned_sw:
            # test_on_flt.py:87 / test_on_crohd.py:133 pass their Summ_writer; every log_freq-th sample has save_this set.
            # The reference only draws there -- the numeric outputs do not depend on it -- so the evaluation must go on.
            import warnings
            warnings.warn("pips_amd.Pips: sw.save_this is set; the tensorboard drawings of nets/pips.py:477-598 are not "
                          "produced (outputs are unaffected)", stacklevel=2)
            self._warned_sw = True
        B, N, D = xys.shape
        assert D == 2
        B2, S, C3, H, W = rgbs.shape
        assert B2 == B and C3 == 3 and S == self.S
        if not rgbs.is_cuda:
            raise _lib.PipsHipError("pips_amd.Pips needs CUDA/HIP tensors (the reference itself calls .cuda(), "
                                    "nets/pips.py:429); there is no CPU fallback")
        lib = _lib.load()
        dev = rgbs.device
        f32 = torch.float32
        u8 = rgbs.dtype == torch.uint8               # decoded frames (demo.py:136-144) are read as they are
        rgbs_c = rgbs.contiguous() if u8 else rgbs.contiguous().to(f32)
        xys_c = xys.to(dev).contiguous().to(f32)
        ci = None if coords_init is None else coords_init.to(dev).contiguous().to(f32)
        fi = None if feat_init is None else feat_init.to(dev).contiguous().to(f32)
        if ci is not None:
            assert tuple(ci.shape) == (B, S, N, 2)
        if fi is not None:
            assert tuple(fi.shape) == (B, N, self.latent_dim)
        with torch.cuda.device(dev):
            arena, times = self._aux(dev)
            ws = self._workspace(lib, (B, S, H, W, N, int(self.stride)), dev)
            trajs = torch.empty(iters + 1, B, S, N, 2, dtype=f32, device=dev)
            vis_e = torch.empty(B, S, N, dtype=f32, device=dev)
            ffeat = torch.empty(B, N, self.latent_dim, dtype=f32, device=dev)
            # evaluation call (test_on_flt.py:87): the score-map loss terms of every iteration come out of the same
            # forward -- heat maps are correlated and reduced on the fly, the (B,S,N,H8,W8) volume is never stored
            ce_tgt = ce_terms = ce_ws = None
            H8, W8 = H // int(self.stride), W // int(self.stride)
            if trajs_g is not None and iters > 0:
                ce_tgt = _score_map_targets(trajs_g.to(dev), vis_g.to(dev), valids.to(dev), float(self.stride), H8, W8)
                ce_terms = torch.empty(iters, B * N * S, 2, dtype=f32, device=dev)
                ce_ws = torch.empty(lib.pips_score_map_workspace_bytes(B, S, H8, W8) // 4, dtype=f32, device=dev)
            rc = lib.pips_forward_ce(_lib.ptr(arena), _lib.ptr(rgbs_c), _lib.ptr(xys_c), _lib.ptr(ci), _lib.ptr(fi),
                                     _lib.ptr(times), B, S, H, W, N, int(self.stride), int(iters),
                                     self._flags() | (8 if u8 else 0),
                                     _lib.ptr(ws), ws.numel() * 4, _lib.ptr(trajs), _lib.ptr(vis_e), _lib.ptr(ffeat),
                                     _lib.ptr(ce_tgt), _lib.ptr(ce_terms), _lib.ptr(ce_ws),
                                     0 if ce_ws is None else ce_ws.numel() * 4,
                                     C.c_void_p(torch.cuda.current_stream().cuda_stream))
            _lib.check(rc, "pips_forward")
        coord_predictions = [trajs[i + 1] for i in range(iters)]
        # nets/pips.py:474-475,539,562-563: two copies of the start, every iterate, two of the end
        coord_predictions2 = [trajs[0], trajs[0]] + coord_predictions + [trajs[iters], trajs[iters]]
        losses = None
        if trajs_g is not None:
            losses = _inference_losses(coord_predictions, vis_e, trajs_g, vis_g, valids, ce_tgt=ce_tgt, ce_terms=ce_terms,
                                       npix=H8 * W8)
        if return_feat:
            return coord_predictions, coord_predictions2, vis_e, ffeat, losses
        return coord_predictions, coord_predictions2, vis_e, losses


    # ------------------------------------------------------------------ encoder / tracker split
    @torch.no_grad()
    def encode(self, rgbs, frames_per_pass: int = 16) -> FeatureCache:
        """BasicEncoder + pyramid of every frame of ``rgbs (B,T,3,H,W)`` (0..255), once.
        Replaces the per-chunk / per-hop encoder re-runs of test_on_davis.py:116-118 and
        chain_demo.py:54.  Frames are encoded ``frames_per_pass`` at a time to bound the
        activation workspace for long videos."""
        if not rgbs.is_cuda:
            raise _lib.PipsHipError("pips_amd.Pips needs CUDA/HIP tensors; there is no CPU fallback")
        lib = _lib.load()
        B, T, C3, H, W = rgbs.shape
        assert C3 == 3
        dev, st = rgbs.device, int(self.stride)
        F = B * T
        with torch.cuda.device(dev):
            arena = self._aux(dev)[0]
            frames = (rgbs.contiguous() if rgbs.dtype == torch.uint8 else rgbs.contiguous().to(torch.float32))
            frames = frames.reshape(F, 3, H, W)
            eb = bool(self._flags() & 4)
            sp = bool(self._flags() & 16)
            if F <= frames_per_pass:
                pyr = ops.encoder_fwd(arena, frames, st, bf16=eb, split=sp)
            else:
                pyr = torch.empty(lib.pips_pyramid_floats(F, H, W, st), dtype=torch.float32, device=dev)
                dst = ops.pyramid_levels(pyr, F, H, W, st)
                for f0 in range(0, F, frames_per_pass):
                    f1 = min(F, f0 + frames_per_pass)
                    part = ops.encoder_fwd(arena, frames[f0:f1], st, bf16=eb, split=sp)
                    for d, p in zip(dst, ops.pyramid_levels(part, f1 - f0, H, W, st)):
                        d[f0:f1].copy_(p)
                if eb:          # the parts' mirrors were laid out for their own frame counts: rewrite the whole one
                    _lib.check(lib.pips_pyramid_mirror(_lib.ptr(pyr), F, H, W, st, C.c_void_p(torch.cuda.current_stream().cuda_stream)),
                               "pips_pyramid_mirror")
        return FeatureCache(pyr, B, T, H, W, st, bf16_maps=eb and not sp)

    @torch.no_grad()
    def track(self, cache: FeatureCache, xys, coords_init=None, feat_init=None, iters=3, win_start=None,
              return_feat=False):
        """The update loop of ``forward`` (nets/pips.py:450-563) on cached maps.  ``win_start``
        ``(B,N)`` int = first frame of each particle's 8-frame window inside the ``T`` cached
        frames (default 0); frames past the end repeat the last one (chain_demo.py:50-52).
        Returns the same tuple as ``forward`` (losses = None)."""
        lib = _lib.load()
        B, N, D = xys.shape
        assert D == 2 and B == cache.B
        dev, f32, S = cache.pyr.device, torch.float32, self.S
        H8, W8 = cache.map_size
        xys_c = xys.to(dev).contiguous().to(f32)
        ci = None if coords_init is None else coords_init.to(dev).contiguous().to(f32)
        fi = None if feat_init is None else feat_init.to(dev).contiguous().to(f32)
        ws_i = None if win_start is None else win_start.to(dev).contiguous().to(torch.int32)
        if ws_i is not None:
            assert tuple(ws_i.shape) == (B, N)
        elif cache.T != S:
            ws_i = torch.zeros(B, N, dtype=torch.int32, device=dev)
        with torch.cuda.device(dev):
            arena, times = self._aux(dev)
            fl = self._flags()
            if cache.bf16_maps and (fl & 2) and not (fl & 16):
                fl |= 32        # PIPS_FLAG_BF16_MAPS: bf16 mixer on maps of the bf16 encoder -> the gather reads their bf16 mirror
            nb = lib.pips_track_workspace_bytes_s(B, N, S)
            # ONE tracker workspace per (device, stream), grown on demand: chained tracking calls this with
            # a different (shrinking) N at every hop
            key = ("track", str(dev), int(torch.cuda.current_stream(dev).cuda_stream))
            with self._lock:
                ws = self._ws.get(key)
                if ws is None or ws.numel() * 4 < nb:
                    ws = self._ws[key] = torch.empty(nb // 4, dtype=f32, device=dev)
            trajs = torch.empty(iters + 1, B, S, N, 2, dtype=f32, device=dev)
            vis_e = torch.empty(B, S, N, dtype=f32, device=dev)
            ffeat = torch.empty(B, N, self.latent_dim, dtype=f32, device=dev)
            rc = lib.pips_track_s(_lib.ptr(arena), _lib.ptr(cache.pyr), B, cache.T, H8, W8, _lib.ptr(xys_c), _lib.ptr(ci),
                                  _lib.ptr(fi), _lib.ptr(ws_i), _lib.ptr(times), N, int(cache.stride), int(iters),
                                  fl, S, _lib.ptr(ws), ws.numel() * 4, _lib.ptr(trajs), _lib.ptr(vis_e),
                                  _lib.ptr(ffeat), None, None, None, 0, C.c_void_p(torch.cuda.current_stream().cuda_stream))
            _lib.check(rc, "pips_track_s")
        preds = [trajs[i + 1] for i in range(iters)]
        preds2 = [trajs[0], trajs[0]] + preds + [trajs[iters], trajs[iters]]
        if return_feat:
            return preds, preds2, vis_e, ffeat, None
        return preds, preds2, vis_e, None


def _masked_mean(x, mask):
    return (x * mask).sum() / (mask.sum() + 1e-6)           # utils.basic.reduce_masked_mean (EPS = 1e-6)


def balanced_ce_loss(pred, gt, valid=None):
    """nets/pips.py:14-37: (balanced_loss, per-element loss).  Evaluation-side torch code on a few KB of outputs."""
    assert pred.shape == gt.shape and (valid is None or valid.shape == gt.shape)
    if valid is None:
        valid = torch.ones_like(gt)
    pos = (gt > 0.95).float()
    neg = (gt < 0.05).float()
    a = -(pos * 2.0 - 1.0) * pred
    b = torch.relu(a)
    loss = b + torch.log(torch.exp(-b) + torch.exp(a - b))
    return _masked_mean(loss, pos * valid) + _masked_mean(loss, neg * valid), loss


def sequence_loss(flow_preds, flow_gt, vis, valids, gamma=0.8):
    """nets/pips.py:39-56: exponentially weighted L1 over the iterates."""
    n = len(flow_preds)
    loss = 0.0
    for i, p in enumerate(flow_preds):
        loss = loss + gamma ** (n - i - 1) * _masked_mean((p - flow_gt).abs().mean(dim=3), valids)
    return loss / n


def _score_map_targets(trajs_g, vis_g, valids, stride, H8, W8):
    """score_map_loss's target selection (nets/pips.py:62-69) as the (B*N*S, 3) table pips_forward_ce takes: rounded
    target pixel (torch.round: half to even) in map coordinates and whether the heat map is used."""
    B, S, N, _ = trajs_g.shape
    xy = (trajs_g.float() / stride).round()
    x, y = xy[..., 0], xy[..., 1]
    use = (x >= 0) & (x <= W8 - 1) & (y >= 0) & (y <= H8 - 1) & (valids > 0) & (vis_g > 0)
    t = torch.stack([x.clamp(0, W8 - 1), y.clamp(0, H8 - 1), use.float()], dim=-1)              # (B,S,N,3)
    return t.permute(0, 2, 1, 3).reshape(B * N * S, 3).contiguous()


def _inference_losses(preds, vis_e, trajs_g, vis_g, valids, gamma=0.8, ce_tgt=None, ce_terms=None, npix=0):
    """The losses tuple of nets/pips.py:600-606 on the outputs (evaluation scripts pass trajs_g, test_on_flt.py:87):
    (seq_loss, vis_loss, ce_loss).  ce_loss = score_map_loss (:58-92) from the per-row terms of pips_forward_ce: the
    balanced cross-entropy's two masked means (utils.basic.reduce_masked_mean, EPS 1e-6) over the used heat maps of all
    iterations -- one positive pixel each, npix - 1 negatives."""
    vis_loss = balanced_ce_loss(vis_e, vis_g, valids)[0]
    if len(preds) == 0:
        return torch.zeros((), device=vis_e.device), vis_loss, None
    ce_loss = None
    if ce_terms is not None:
        n_maps = ce_tgt[:, 2].sum() * ce_terms.shape[0]
        t = ce_terms.double().sum(dim=(0, 1))
        ce_loss = (t[0] / (n_maps + 1e-6) + t[1] / (n_maps * (npix - 1) + 1e-6)).float()
    return sequence_loss(preds, trajs_g, vis_g, valids, gamma), vis_loss, ce_loss
