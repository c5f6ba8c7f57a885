"""Parameter table of the PIPs tracker and the repack into the layouts the HIP kernels read.

The drop-in boundary includes the state dict: the reference checkpoint
(`saverloader.py:58-59`, ``load_state_dict(..., strict=False)``) must load into our
``Pips`` without a single renamed key, so the table below reproduces the 200
names/shapes that ``nets/pips.py:400-426`` creates (encoder `:184-244`, mixer
`:111-123`, heads `:416-426`).  Nothing here computes on the hot path; it only
names tensors, draws initial values and re-lays weights out once per load.
"""
from __future__ import annotations

import math
from collections import OrderedDict

import torch

LATENT = 128          # nets/pips.py:408
MIX_DIM = 512         # nets/pips.py:298
MIX_DEPTH = 12        # nets/pips.py:300
CORR_LEVELS = 4       # nets/pips.py:409
CORR_RADIUS = 3       # nets/pips.py:410
KITCHEN = CORR_LEVELS * (2 * CORR_RADIUS + 1) ** 2 + LATENT + 64 * 3 + 3   # 519, nets/pips.py:289
KITCHEN_PAD = 544     # K of the first mixer Linear zero-padded to a multiple of 32 (PIPS_KIN_PAD)


def encoder_convs():
    """(key prefix, cout, cin, k, stride, pad) for the 22 convs, in execution order
    (nets/pips.py:206-223, 135-136, 169-170)."""
    out = [("fnet.conv1", 64, 3, 7, 2, 3)]
    cin = 64
    for li, (dim, st) in enumerate(((64, 1), (96, 2), (128, 2), (128, 2)), start=1):
        for bi in range(2):
            s = st if bi == 0 else 1
            p = f"fnet.layer{li}.{bi}"
            out.append((p + ".conv1", dim, cin, 3, s, 1))
            out.append((p + ".conv2", dim, dim, 3, 1, 1))
            if s != 1:
                out.append((p + ".downsample.0", dim, cin, 1, s, 0))
            cin = dim
    out.append(("fnet.conv2", 256, 64 + 96 + 128 + 128, 3, 1, 1))
    out.append(("fnet.conv3", 128, 256, 1, 1, 0))
    return out


def param_table(S: int = 8):
    """Ordered ``name -> (shape, kind)``; kind selects the initial distribution."""
    t = OrderedDict()
    for p, co, ci, k, _s, _pad in encoder_convs():
        t[p + ".weight"] = ((co, ci, k, k), "conv_w")
        t[p + ".bias"] = ((co,), ("fan_in_u", ci * k * k))
    md = "delta_block.to_delta"
    t[f"{md}.0.weight"] = ((MIX_DIM, KITCHEN), ("fan_in_u", KITCHEN))
    t[f"{md}.0.bias"] = ((MIX_DIM,), ("fan_in_u", KITCHEN))
    for d in range(1, MIX_DEPTH + 1):
        # token-mixing MLP over the S axis, stored as Conv1d(k=1) (nets/pips.py:112,117)
        t[f"{md}.{d}.0.fn.0.weight"] = ((4 * S, S, 1), ("fan_in_u", S))
        t[f"{md}.{d}.0.fn.0.bias"] = ((4 * S,), ("fan_in_u", S))
        t[f"{md}.{d}.0.fn.3.weight"] = ((S, 4 * S, 1), ("fan_in_u", 4 * S))
        t[f"{md}.{d}.0.fn.3.bias"] = ((S,), ("fan_in_u", 4 * S))
        t[f"{md}.{d}.0.norm.weight"] = ((MIX_DIM,), "ones")
        t[f"{md}.{d}.0.norm.bias"] = ((MIX_DIM,), "zeros")
        # channel-mixing MLP (nets/pips.py:118)
        t[f"{md}.{d}.1.fn.0.weight"] = ((4 * MIX_DIM, MIX_DIM), ("fan_in_u", MIX_DIM))
        t[f"{md}.{d}.1.fn.0.bias"] = ((4 * MIX_DIM,), ("fan_in_u", MIX_DIM))
        t[f"{md}.{d}.1.fn.3.weight"] = ((MIX_DIM, 4 * MIX_DIM), ("fan_in_u", 4 * MIX_DIM))
        t[f"{md}.{d}.1.fn.3.bias"] = ((MIX_DIM,), ("fan_in_u", 4 * MIX_DIM))
        t[f"{md}.{d}.1.norm.weight"] = ((MIX_DIM,), "ones")
        t[f"{md}.{d}.1.norm.bias"] = ((MIX_DIM,), "zeros")
    t[f"{md}.{MIX_DEPTH + 1}.weight"] = ((MIX_DIM,), "ones")
    t[f"{md}.{MIX_DEPTH + 1}.bias"] = ((MIX_DIM,), "zeros")
    t[f"{md}.{MIX_DEPTH + 3}.weight"] = ((S * (LATENT + 2), MIX_DIM), ("fan_in_u", MIX_DIM))
    t[f"{md}.{MIX_DEPTH + 3}.bias"] = ((S * (LATENT + 2),), ("fan_in_u", MIX_DIM))
    t["norm.weight"] = ((LATENT,), "ones")
    t["norm.bias"] = ((LATENT,), "zeros")
    t["ffeat_updater.0.weight"] = ((LATENT, LATENT), ("fan_in_u", LATENT))
    t["ffeat_updater.0.bias"] = ((LATENT,), ("fan_in_u", LATENT))
    t["vis_predictor.0.weight"] = ((1, LATENT), ("fan_in_u", LATENT))
    t["vis_predictor.0.bias"] = ((1,), ("fan_in_u", LATENT))
    return t


def init_state_dict(seed: int = 0, S: int = 8, tamed: bool = False, dtype=torch.float32):
    """Seeded random weights with the reference's distributions: Kaiming-normal
    (fan_out, relu) conv weights (nets/pips.py:229-231), U(+-1/sqrt(fan_in)) for every
    Linear/Conv1d tensor and conv bias (torch defaults), ones/zeros for the affine norms.

    One generator per tensor, keyed by position, so a tensor's values do not depend on
    construction order.  ``tamed`` scales the last mixer Linear by 0.02:
    with untrained weights the update map is chaotic, which makes end-to-end I=6
    comparisons undecidable at 1e-3 px; the tamed set keeps displacements sub-pixel.
    """
    sd = OrderedDict()
    for i, (name, (shape, kind)) in enumerate(param_table(S).items()):
        g = torch.Generator().manual_seed(seed * 100003 + i)
        if kind == "ones":
            w = torch.ones(shape, dtype=dtype)
        elif kind == "zeros":
            w = torch.zeros(shape, dtype=dtype)
        elif kind == "conv_w":
            co, _ci, kh, kw = shape
            std = math.sqrt(2.0 / (co * kh * kw))
            w = (torch.randn(shape, generator=g, dtype=torch.float32) * std).to(dtype)
        else:
            bound = 1.0 / math.sqrt(kind[1])
            w = ((torch.rand(shape, generator=g, dtype=torch.float32) * 2 - 1) * bound).to(dtype)
        sd[name] = w
    if tamed:
        k = f"delta_block.to_delta.{MIX_DEPTH + 3}"
        sd[k + ".weight"] = sd[k + ".weight"] * 0.02
        sd[k + ".bias"] = sd[k + ".bias"] * 0.02
    return sd
