"""Golden-vector case table shared by make_golden.py (which runs the REAL reference in the
build container) and the tests (which regenerate the seeded inputs and compare).

Inputs and weights are regenerated from seeds (torch CPU generators are deterministic for
a fixed torch build; the GPU box runs the same image), so only the reference's OUTPUTS
are stored.  The one real-image case stores its frames (uint8, half resolution)."""
from __future__ import annotations

import os

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))

# name: (B, N, H, W, stride, iters, tamed, special)
CASES = {
    "s8_raw_i3": dict(B=1, N=16, H=128, W=160, stride=8, iters=3, tamed=False, border=True),
    "s8_tamed_i6": dict(B=2, N=5, H=136, W=200, stride=8, iters=6, tamed=True, border=False),
    "s4_raw_i2": dict(B=1, N=16, H=96, W=128, stride=4, iters=2, tamed=False, border=True),
    "s8_init_i2": dict(B=1, N=7, H=128, W=160, stride=8, iters=2, tamed=True, border=False, init=True),
    "demo_half_s4_i2": dict(B=1, N=16, H=180, W=320, stride=4, iters=2, tamed=True, border=False, demo=True),
    # BASELINE configs[0] at its OWN size: the first 8 demo_images as demo.py feeds them (640x360, no resize needed),
    # Pips(stride=4), the 4x4 grid of demo.py:32-36, I = 2
    "demo_full_s4_i2": dict(B=1, N=16, H=360, W=640, stride=4, iters=2, tamed=True, border=False, demo="full"),
}

# Pips(S != 8): the reference sizes the token-mixing weights and the head by S (nets/pips.py:295-301, 401-402).  An odd S
# (head rows 650 = not a multiple of 4), one beyond 8 (two row groups in the state update) and a short one at stride 4.
WINDOW_CASES = {
    "w5_tamed_i3": dict(S=5, B=1, N=6, H=128, W=160, stride=8, iters=3, tamed=True, border=False),
    "w12_tamed_i2": dict(S=12, B=1, N=5, H=128, W=160, stride=8, iters=2, tamed=True, border=False),
    "w4_raw_s4_i2": dict(S=4, B=2, N=7, H=96, W=128, stride=4, iters=2, tamed=False, border=True),
    "w24_tamed_i2": dict(S=24, B=1, N=4, H=128, W=160, stride=8, iters=2, tamed=True, border=False),     # beyond 16: the SMAX = 32 kernels
}


def make_inputs(case: dict, seed: int = 1, S: int = 8):
    """(xys, rgbs, coords_init, feat_init) on CPU, fp32.  The window length is case["S"] when the case names one."""
    S = case.get("S", S)
    B, N, H, W = case["B"], case["N"], case["H"], case["W"]
    g = torch.Generator().manual_seed(seed)
    if case.get("demo"):
        # (8,180,320,3) uint8, or the frames at their native (8,360,640,3)
        frames = np.load(os.path.join(HERE, "demo_full_frames.npz" if case["demo"] == "full" else "demo_half_frames.npz"))["frames"]
        rgbs = torch.from_numpy(frames).permute(0, 3, 1, 2).float().unsqueeze(0)
        # demo.py:32-36: uniform sqrt(N) x sqrt(N) grid with an 8 px margin
        n_ = int(round(N ** 0.5))
        gy, gx = torch.meshgrid(torch.linspace(8, H - 8, n_), torch.linspace(8, W - 8, n_), indexing="ij")
        xys = torch.stack([gx.reshape(-1), gy.reshape(-1)], dim=-1).unsqueeze(0)
    else:
        rgbs = torch.randint(0, 256, (B, S, 3, H, W), generator=g).float()
        xys = torch.rand(B, N, 2, generator=g) * torch.tensor([W - 1.0, H - 1.0])
        if case.get("border"):
            # exercise zero-padded window taps, clamped point samples and integer coordinates
            xys[0, 0] = torch.tensor([0.0, 0.0])
            xys[0, 1] = torch.tensor([W - 1.0, H - 1.0])
            xys[0, 2] = torch.tensor([3.0, H - 2.5])
            xys[0, 3] = torch.tensor([W - 1.25, 1.0])
            xys[0, 4] = torch.tensor([16.0, 24.0])
    coords_init = feat_init = None
    if case.get("init"):
        coords_init = xys.reshape(B, 1, N, 2).repeat(1, S, 1, 1) + torch.randn(B, S, N, 2, generator=g) * 2.0
        feat_init = torch.randn(B, N, 128, generator=g) * 0.5
    return xys, rgbs, coords_init, feat_init


def make_targets(case: dict, seed: int = 7, S: int = 8):
    """Ground-truth-like targets for the losses of nets/pips.py:600-606: (trajs_g (B,S,N,2) px, vis_g, valids (B,S,N))."""
    S = case.get("S", S)
    B, N = case["B"], case["N"]
    xys = make_inputs(case)[0]
    g = torch.Generator().manual_seed(seed)
    trajs_g = xys.reshape(B, 1, N, 2).repeat(1, S, 1, 1) + torch.randn(B, S, N, 2, generator=g) * 3.0
    vis_g = (torch.rand(B, S, N, generator=g) > 0.3).float()
    valids = (torch.rand(B, S, N, generator=g) > 0.1).float()
    return trajs_g, vis_g, valids


# chained long-video case (chain_demo.py:39-83): a slowly changing 13-frame video, three queries
CHAIN_CASE = dict(T=13, H=128, W=160, N=3, stride=8, iters=6, tamed=True)


def make_chain_inputs(case: dict = CHAIN_CASE, seed: int = 11):
    """(video (1,T,3,H,W) 0..255, xy0 (1,N,2) px).  The level-3 map is 2x2: a 1-pixel level divides by (W-1)=0
    (nets/pips.py:318) and the reference's threshold scan would never end."""
    g = torch.Generator().manual_seed(seed)
    T, H, W, N = case["T"], case["H"], case["W"], case["N"]
    base = torch.randint(0, 256, (1, 1, 3, H, W), generator=g).float()
    video = torch.cat([(base * (1 - 0.04 * t) + 9.0 * t).clamp(0, 255).round() for t in range(T)], dim=1)
    xy0 = torch.rand(1, N, 2, generator=g) * torch.tensor([W - 17.0, H - 17.0]) + 8.0
    return video, xy0
