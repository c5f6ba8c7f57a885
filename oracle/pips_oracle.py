"""CPU ORACLE for the PIPs inference hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this file.  ``pips_amd`` never does: its forward fails loudly when the HIP
library is missing.

What it is: a staged, functional restatement (plain torch ops on CPU, fp32 or fp64) of
``nets.pips.Pips.forward`` in inference mode, taking the reference's own 200-key state
dict, and exposing every intermediate the HIP kernels are checked against ("taps").
Each function cites the reference lines it follows.  The arithmetic primitives
(conv2d, instance_norm, interpolate, grid_sample, layer_norm, gelu, matmul) are the
same ATen ops the reference executes, so the restatement is the reference's algorithm
on the reference's numerics; ``oracle/check_against_reference.py`` pins it by running
the unmodified ``/root/reference/nets/pips.py`` on the same weights and inputs (max
abs difference 0.0 on every fixture at the time of writing), and
``tests/golden/`` holds outputs produced by that reference run.

Parity status: the reference repo holds NO tests, golden vectors or known-answer
fixtures for this path, so parity is pinned only by outputs of the
reference itself executed in the build container (tests/golden/make_golden.py).
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

S_FRAMES = 8
LATENT = 128
LEVELS = 4
RADIUS = 3
MIX_DEPTH = 12


# --------------------------------------------------------------------------- encoder
def _inorm(x):
    # nn.InstanceNorm2d(affine=False, track_running_stats=False, eps=1e-5): nets/pips.py:153-157,199-201
    return F.instance_norm(x, eps=1e-5)


def _conv(sd, key, x, stride, pad):
    return F.conv2d(x, sd[key + ".weight"], sd[key + ".bias"], stride=stride, padding=pad)


def _res_block(sd, p, x, stride):
    """ResidualBlock.forward, nets/pips.py:173-181."""
    y = F.relu(_inorm(_conv(sd, p + ".conv1", x, stride, 1)))
    y = F.relu(_inorm(_conv(sd, p + ".conv2", y, 1, 1)))
    if stride != 1:
        x = _inorm(_conv(sd, p + ".downsample.0", x, stride, 0))   # nets/pips.py:169-170
    return F.relu(x + y)


def encoder(sd, x, stride, taps=None):
    """BasicEncoder.forward (instance norm, non-shallow), nets/pips.py:247-281.
    x: (F,3,H,W) already scaled to [-1,1]."""
    _, _, H, W = x.shape
    x = F.relu(_inorm(_conv(sd, "fnet.conv1", x, 2, 3)))                      # :251-253
    if taps is not None:
        taps["enc_stem"] = x
    outs = []
    for li, st in ((1, 1), (2, 2), (3, 2), (4, 2)):                           # :265-268
        x = _res_block(sd, f"fnet.layer{li}.0", x, st)
        x = _res_block(sd, f"fnet.layer{li}.1", x, 1)
        outs.append(x)
        if taps is not None:
            taps[f"enc_layer{li}"] = x
    size = (H // stride, W // stride)
    outs = [F.interpolate(o, size, mode="bilinear", align_corners=True) for o in outs]   # :269-272
    x = torch.cat(outs, dim=1)                                                 # :273 (a,b,c,d)
    if taps is not None:
        taps["enc_cat"] = x
    x = F.relu(_inorm(_conv(sd, "fnet.conv2", x, 1, 1)))                       # :273-275
    x = _conv(sd, "fnet.conv3", x, 1, 0)                                       # :276
    return x


# ------------------------------------------------------------------- correlation side
def build_pyramid(fmaps):
    """CorrBlock.__init__, nets/pips.py:346-352: 3x avg_pool2d(2, stride 2), floor sizes."""
    B, S, C, H, W = fmaps.shape
    pyr = [fmaps]
    for _ in range(LEVELS - 1):
        f = F.avg_pool2d(pyr[-1].reshape(B * S, C, H, W), 2, stride=2)
        H, W = f.shape[-2:]
        pyr.append(f.reshape(B, S, C, H, W))
    return pyr


def point_sample(im, x, y):
    """utils.samp.bilinear_sample2d, utils/samp.py:5-78: the four neighbour INDICES are
    clamped to the border, the weights are not (samp.py:21-29 vs 59-62).
    im (B,C,H,W); x,y (B,N) -> (B,N,C)."""
    B, C, H, W = im.shape
    x0 = torch.floor(x)
    y0 = torch.floor(y)
    x1 = x0 + 1
    y1 = y0 + 1
    xi0 = x0.long().clamp(0, W - 1)
    xi1 = x1.long().clamp(0, W - 1)
    yi0 = y0.long().clamp(0, H - 1)
    yi1 = y1.long().clamp(0, H - 1)
    flat = im.permute(0, 2, 3, 1).reshape(B, H * W, C)

    def take(yi, xi):
        idx = (yi * W + xi).unsqueeze(-1).expand(B, -1, C)
        return torch.gather(flat, 1, idx)

    w00 = ((x1 - x) * (y1 - y)).unsqueeze(2)
    w01 = ((x - x0) * (y1 - y)).unsqueeze(2)
    w10 = ((x1 - x) * (y - y0)).unsqueeze(2)
    w11 = ((x - x0) * (y - y0)).unsqueeze(2)
    # summation order of samp.py:64-65
    return w00 * take(yi0, xi0) + w01 * take(yi0, xi1) + w10 * take(yi1, xi0) + w11 * take(yi1, xi1)


def corr_sample(pyramid, ffeats, coords):
    """CorrBlock.corr + CorrBlock.sample (nets/pips.py:384-398, 355-382) with
    bilinear_sampler (:313-328).  ffeats (B,S,N,C), coords (B,S,N,2) in stride-px.
    Returns (B,S,N,4*49): per level the 7x7 window in the reference's TRANSPOSED
    order -- entry k=i*7+j samples x=cx+(i-3), y=cy+(j-3) (delta is stacked (dy,dx) but
    added to (x,y), :369-375)."""
    B, S, N, C = ffeats.shape
    r = RADIUS
    d = torch.linspace(-r, r, 2 * r + 1, dtype=coords.dtype, device=coords.device)
    delta = torch.stack(torch.meshgrid(d, d, indexing="ij"), dim=-1)          # :369-371
    outs = []
    for lvl, fm in enumerate(pyramid):
        _, _, _, H, W = fm.shape
        corr = torch.matmul(ffeats, fm.reshape(B, S, C, H * W))               # :394-395
        corr = corr / torch.sqrt(torch.tensor(float(C), dtype=corr.dtype, device=corr.device))    # :397
        centroid = coords.reshape(B * S * N, 1, 1, 2) / 2 ** lvl              # :373
        pts = centroid + delta.view(1, 2 * r + 1, 2 * r + 1, 2)               # :375
        xg = 2 * pts[..., 0:1] / (W - 1) - 1                                  # :318
        yg = 2 * pts[..., 1:2] / (H - 1) - 1                                  # :319
        grid = torch.cat([xg, yg], dim=-1)
        smp = F.grid_sample(corr.reshape(B * S * N, 1, H, W), grid, mode="bilinear",
                            padding_mode="zeros", align_corners=True)         # :322
        outs.append(smp.view(B, S, N, -1))
    return torch.cat(outs, dim=-1)                                             # :381


# --------------------------------------------------------------------------- mixer side
def embed3d(xyz, C=64):
    """utils.misc.get_3d_embedding, utils/misc.py:44-69.  (M,S,3) -> (M,S,3*C+3).
    The frequency table is float32 in the reference regardless of input dtype."""
    div = (torch.arange(0, C, 2, dtype=torch.float32, device=xyz.device) * (1000.0 / C)).to(xyz.dtype).reshape(1, 1, C // 2)
    parts = []
    for a in range(3):
        v = xyz[:, :, a:a + 1] * div
        pe = torch.stack([torch.sin(v), torch.cos(v)], dim=-1).reshape(*v.shape[:2], C)   # interleave
        parts.append(pe)
    return torch.cat(parts + [xyz], dim=2)


def mixer_input(ffeats, fcorrs, coords):
    """nets/pips.py:517-522 + DeltaBlock.forward :304-308: per particle (B*N) an
    (S, 519) token matrix [ffeat | fcorr | sincos(192) | dx dy t]."""
    B, S, N, C = ffeats.shape
    fc = fcorrs.permute(0, 2, 1, 3).reshape(B * N, S, -1)
    fl = (coords - coords[:, 0:1]).permute(0, 2, 1, 3).reshape(B * N, S, 2)
    t = torch.linspace(0, S, S, dtype=coords.dtype, device=coords.device).reshape(1, S, 1).repeat(B * N, 1, 1)   # :519 (0..S, not 0..S-1)
    fl = torch.cat([fl, t], dim=2)
    ff = ffeats.permute(0, 2, 1, 3).reshape(B * N, S, C)
    return torch.cat([ff, fc, embed3d(fl)], dim=2)


def mixer(sd, x, taps=None):
    """MLPMixer, nets/pips.py:111-123 (PreNormResidual :93-100, FeedForward :102-109).
    x (M,S,519) -> (M, S*130)."""
    md = "delta_block.to_delta"
    x = F.linear(x, sd[f"{md}.0.weight"], sd[f"{md}.0.bias"])
    if taps is not None:
        taps["mix_in_proj"] = x
    D = x.shape[-1]
    for d in range(1, MIX_DEPTH + 1):
        h = F.layer_norm(x, (D,), sd[f"{md}.{d}.0.norm.weight"], sd[f"{md}.{d}.0.norm.bias"])
        h = F.conv1d(h, sd[f"{md}.{d}.0.fn.0.weight"], sd[f"{md}.{d}.0.fn.0.bias"])     # tokens = conv channels
        h = F.gelu(h)
        h = F.conv1d(h, sd[f"{md}.{d}.0.fn.3.weight"], sd[f"{md}.{d}.0.fn.3.bias"])
        x = h + x
        h = F.layer_norm(x, (D,), sd[f"{md}.{d}.1.norm.weight"], sd[f"{md}.{d}.1.norm.bias"])
        h = F.linear(h, sd[f"{md}.{d}.1.fn.0.weight"], sd[f"{md}.{d}.1.fn.0.bias"])
        h = F.gelu(h)
        h = F.linear(h, sd[f"{md}.{d}.1.fn.3.weight"], sd[f"{md}.{d}.1.fn.3.bias"])
        x = h + x
        if taps is not None and d == 1:
            taps["mix_block1"] = x
    x = F.layer_norm(x, (D,), sd[f"{md}.{MIX_DEPTH + 1}.weight"], sd[f"{md}.{MIX_DEPTH + 1}.bias"])
    x = x.mean(dim=1)                                                           # Reduce('b n c -> b c','mean')
    return F.linear(x, sd[f"{md}.{MIX_DEPTH + 3}.weight"], sd[f"{md}.{MIX_DEPTH + 3}.bias"])


def update_step(sd, ffeats, coords, coords0, delta):
    """nets/pips.py:525-536.  delta (B*N, S, 130).  Returns new (ffeats, coords)."""
    B, S, N, C = ffeats.shape
    dc = delta[:, :, :2]
    df = delta[:, :, 2:].reshape(B * N * S, C)
    ff = ffeats.permute(0, 2, 1, 3).reshape(B * N * S, C)
    h = F.group_norm(df, 1, sd["norm.weight"], sd["norm.bias"], eps=1e-5)       # GroupNorm(1,C) on 2-D input
    h = F.gelu(F.linear(h, sd["ffeat_updater.0.weight"], sd["ffeat_updater.0.bias"]))
    ff = h + ff
    ffeats = ff.reshape(B, N, S, C).permute(0, 2, 1, 3)
    coords = coords + dc.reshape(B, N, S, 2).permute(0, 2, 1, 3)
    coords = coords.clone()
    coords[:, 0] = coords0[:, 0]                                                 # :535-536 (inference)
    return ffeats, coords


def iteration(sd, pyramid, ffeats, coords, coords0, taps=None):
    """One pass of the loop body nets/pips.py:499-539 (dead fcp branch :504-511 omitted:
    its result is consumed only by score_map_loss and the sw visualisations)."""
    fcorrs = corr_sample(pyramid, ffeats, coords)
    x = mixer_input(ffeats, fcorrs, coords)
    if taps is not None:
        taps["fcorrs"] = fcorrs
        taps["mix_in"] = x
    delta = mixer(sd, x, taps).reshape(-1, ffeats.shape[1], LATENT + 2)          # DeltaBlock.forward: (B*N, self.S, C+2)
    if taps is not None:
        taps["delta"] = delta
    return update_step(sd, ffeats, coords, coords0, delta)


# --------------------------------------------------------------------------- forward
@torch.no_grad()
def forward(sd, xys, rgbs, iters=3, stride=8, coords_init=None, feat_init=None, taps=None,
            fmaps=None):
    """Pips.forward in inference mode (trajs_g=None, sw=None, is_train=False),
    nets/pips.py:428-611.  Returns (coord_predictions, coord_predictions2, vis_e, ffeat0).
    ``taps``: optional dict filled with intermediates (per-iteration entries are lists).
    ``fmaps``: optional precomputed encoder output (B,S,128,H8,W8) to skip the encoder."""
    B, N, _ = xys.shape
    _, S, _, H, W = rgbs.shape
    dt = sd["fnet.conv1.weight"].dtype
    xys = xys.to(dt)
    if fmaps is None:
        x = 2 * (rgbs.to(dt) / 255.0) - 1.0                                      # :436
        fmaps = encoder(sd, x.reshape(B * S, 3, H, W), stride, taps)
        fmaps = fmaps.reshape(B, S, LATENT, H // stride, W // stride)
    xys_ = xys / float(stride)                                                   # :450
    if coords_init is None:
        coords = xys_.reshape(B, 1, N, 2).repeat(1, S, 1, 1)                     # :453
    else:
        coords = coords_init.to(dt) / stride                                     # :455
    pyramid = build_pyramid(fmaps)
    if feat_init is None:
        ffeat0 = point_sample(fmaps[:, 0], coords[:, 0, :, 0], coords[:, 0, :, 1])   # :463
    else:
        ffeat0 = feat_init.to(dt)
    ffeats = ffeat0.unsqueeze(1).repeat(1, S, 1, 1)
    coords0 = coords.clone()
    if taps is not None:
        taps["fmaps"] = fmaps
        taps["pyramid"] = pyramid
        taps["ffeat0"] = ffeat0
        taps["iters"] = []
    preds, preds2 = [], [coords * stride, coords * stride]                        # :474-475
    for _ in range(iters):
        it = {} if taps is not None else None
        if it is not None:
            it["ffeats_in"] = ffeats
            it["coords_in"] = coords
        ffeats, coords = iteration(sd, pyramid, ffeats, coords, coords0, it)
        if it is not None:
            it["ffeats_out"] = ffeats
            it["coords_out"] = coords
            taps["iters"].append(it)
        preds.append(coords * stride)                                            # :538
        preds2.append(coords * stride)
    vis = F.linear(ffeats.reshape(B * S * N, LATENT), sd["vis_predictor.0.weight"],
                   sd["vis_predictor.0.bias"]).reshape(B, S, N)                  # :559
    preds2 += [coords * stride, coords * stride]                                  # :562-563
    return preds, preds2, vis, ffeat0


def to_dtype(sd, dtype):
    return {k: v.to(dtype) for k, v in sd.items()}


# ------------------------------------------------------------------ evaluation losses (nets/pips.py:14-92, 501-511, 600-606)
def dense_score_maps(pyramid, ffeats):
    """The per-iteration dense score map of nets/pips.py:501-511: every level's correlation volume (CorrBlock.corr,
    :384-398) upsampled to the level-0 size (bilinear, align_corners=True) and summed.  ffeats (B,S,N,C) -> (B,S,N,H8,W8)."""
    B, S, N, C = ffeats.shape
    H8, W8 = pyramid[0].shape[-2:]
    fcp = torch.zeros(B, S, N, H8, W8, dtype=ffeats.dtype, device=ffeats.device)                      # :503
    for fm in pyramid:
        _, _, _, H, W = fm.shape
        corr = torch.matmul(ffeats, fm.reshape(B, S, C, H * W))                                        # :394-395
        corr = corr / torch.sqrt(torch.tensor(float(C), dtype=corr.dtype, device=corr.device))       # :397
        up = F.interpolate(corr.reshape(B * S, N, H, W), (H8, W8), mode="bilinear", align_corners=True)   # :508
        fcp = fcp + up.reshape(B, S, N, H8, W8)                                                        # :509
    return fcp


def _masked_mean(x, mask):
    return (x * mask).sum() / (1e-6 + mask.sum())                      # utils.basic.reduce_masked_mean


def balanced_ce_loss(pred, gt, valid=None):
    """nets/pips.py:14-37."""
    if valid is None:
        valid = torch.ones_like(gt)
    pos = (gt > 0.95).to(pred.dtype)
    neg = (gt < 0.05).to(pred.dtype)
    label = pos * 2.0 - 1.0
    a = -label * pred
    b = F.relu(a)
    loss = b + torch.log(torch.exp(-b) + torch.exp(a - b))
    return _masked_mean(loss, pos * valid) + _masked_mean(loss, neg * valid), loss


def score_map_loss(fcps, trajs_g, vis_g, valids):
    """nets/pips.py:58-92.  fcps (B,S,I,N,H8,W8), trajs_g (B,S,N,2) in map pixels."""
    B, S, I, N, H8, W8 = fcps.shape
    fcp_ = fcps.permute(0, 1, 3, 2, 4, 5).reshape(B * S * N, I, H8, W8)
    xy_ = trajs_g.reshape(B * S * N, 2).round().long()
    vis_ = vis_g.reshape(B * S * N)
    valid_ = valids.reshape(B * S * N)
    x_, y_ = xy_[:, 0], xy_[:, 1]
    ind = (x_ >= 0) & (x_ <= (W8 - 1)) & (y_ >= 0) & (y_ <= (H8 - 1)) & (valid_ > 0) & (vis_ > 0)
    fcp_ = fcp_[ind]
    xy_ = xy_[ind]
    gt_ = torch.zeros_like(fcp_)
    for n in range(fcp_.shape[0]):
        gt_[n, :, xy_[n, 1], xy_[n, 0]] = 1
    return balanced_ce_loss(fcp_.reshape(-1), gt_.reshape(-1))[0]


def sequence_loss(flow_preds, flow_gt, vis, valids, gamma=0.8):
    """nets/pips.py:39-56."""
    n = len(flow_preds)
    loss = 0.0
    for i, p in enumerate(flow_preds):
        loss = loss + gamma ** (n - i - 1) * _masked_mean((p - flow_gt).abs().mean(dim=3), valids)
    return loss / n


def losses(sd, xys, rgbs, trajs_g, vis_g, valids, iters=3, stride=8, coords_init=None, feat_init=None):
    """(seq_loss, vis_loss, ce_loss) of nets/pips.py:600-606 for the forward above."""
    taps = {}
    preds, _, vis, _ = forward(sd, xys, rgbs, iters=iters, stride=stride, coords_init=coords_init, feat_init=feat_init,
                               taps=taps)
    fcps = torch.stack([dense_score_maps(taps["pyramid"], it["ffeats_in"]) for it in taps["iters"]], dim=2)
    return (sequence_loss(preds, trajs_g, vis_g, valids), balanced_ce_loss(vis, vis_g, valids)[0],
            score_map_loss(fcps, trajs_g / float(stride), vis_g, valids))
