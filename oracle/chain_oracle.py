"""CPU ORACLE of the visibility-aware chaining loop -- TEST INFRASTRUCTURE (see pips_oracle.py).

Restates chain_demo.run_model (chain_demo.py:40-83; same loop in test_on_badja.py:64-112):
one particle at a time, 8-frame windows padded by repeating the last frame, the encoder
re-run for every window, the first window's initial feature carried as feat_init, the next
window starting at the latest frame whose visibility clears a threshold that starts at 0.9
and drops by 0.02 each time the scan reaches frame 1."""
from __future__ import annotations

import torch

from . import pips_oracle as O


@torch.no_grad()
def chain(sd, rgbs, xy0, iters=6, stride=8, cache_frames=False):
    """rgbs (1,T,3,H,W), xy0 (1,N,2) -> trajs_e (1,T,N,2), list of hop sequences per particle.

    The loop is restated statement by statement and PINNED to the reference's own text: chain_demo.py cannot be
    imported here (cv2 / tensorboardX / imageio are absent), but tests/golden/make_chain_golden.py executes its lines
    39-83 verbatim (only ``device='cuda'`` -> ``'cpu'``) with the unmodified reference model and stores the result
    (tests/golden/chain_t13.npz); tests/test_oracle_golden.py::test_chain_oracle_against_reference_loop_text holds this
    function to it -- trajectories and hop sequence.  ``cache_frames=True`` is a cost
    shortcut for long/large test videos, not part of the reference: every frame goes through the encoder ONCE and the
    windows index the cached maps -- exact in real arithmetic because InstanceNorm is per frame (nets/pips.py:153-157);
    tests/test_oracle_golden.py checks it against the faithful loop."""
    B, T = rgbs.shape[:2]
    N = xy0.shape[1]
    trajs_e = torch.zeros(B, T, N, 2)
    hops = []
    frame_maps = None
    if cache_frames:
        H, W = rgbs.shape[-2:]
        x = 2 * (rgbs[0] / 255.0) - 1.0
        frame_maps = torch.cat([O.encoder(sd, x[t:t + 1], stride) for t in range(T)], dim=0)     # (T,128,H/s,W/s)
    for n in range(N):                                                    # chain_demo.py:40
        cur, done = 0, False
        traj_e = torch.zeros(B, T, 2)
        traj_e[:, 0] = xy0[:, n]
        feat_init = None
        seq = []
        while not done:
            end = cur + 8
            rgb_seq = rgbs[:, cur:end]
            S_local = rgb_seq.shape[1]
            rgb_seq = torch.cat([rgb_seq, rgb_seq[:, -1].unsqueeze(1).repeat(1, 8 - S_local, 1, 1, 1)], dim=1)
            fmaps = None
            if frame_maps is not None:
                idx = torch.arange(cur, cur + 8).clamp(max=T - 1)
                fmaps = frame_maps[idx].unsqueeze(0)
            preds, _, vis, ffeat = O.forward(sd, traj_e[:, cur].reshape(1, -1, 2), rgb_seq, iters=iters,
                                             stride=stride, feat_init=feat_init, fmaps=fmaps)
            feat_init = ffeat                                             # :57
            vis = torch.sigmoid(vis)
            assert torch.isfinite(vis).all(), "non-finite visibility: the reference's threshold scan (:63-77) would never end"
            xys = preds[-1].reshape(1, 8, 2)
            traj_e[:, cur:end] = xys[:, :S_local]
            thr, si = 0.9, 7                                              # :63-77
            while True:
                if vis[0, si] > thr:
                    break
                si -= 1
                if si == 1:
                    thr -= 0.02
                    si = 7
            seq.append(si)
            cur += si
            done = cur >= T
        trajs_e[:, :, n] = traj_e
        hops.append(seq)
    return trajs_e, hops


@torch.no_grad()
def chain_lockstep(sd, rgbs, xy0, iters=6, stride=8):
    """The same loop as ``chain`` (chain_demo.py:40-83) with all particles advanced side by side, for videos at the size
    of BASELINE configs[4] (T = 100, N = 256) where ``chain`` -- one oracle forward per particle and hop -- takes hours.

    Every live particle is ONE CLIP of the oracle forward (batch row = particle, N = 1): its own 8-frame window of the
    cached per-frame maps (``cache_frames`` of ``chain``: exact, InstanceNorm is per frame), its own start position and
    carried ``feat_init``; nothing in ``pips_oracle.forward`` mixes batch rows, so each row computes what ``chain``
    computes for that particle.  The threshold scan of :63-77 runs per particle in Python, as written.  Runs on
    whatever device ``sd`` / ``rgbs`` live on.  tests/test_oracle_golden.py::test_chain_lockstep_equals_chain holds it to
    ``chain`` (identical hop sequences, trajectories to fp32 round-off of the batched ATen ops)."""
    B, T = rgbs.shape[:2]
    assert B == 1
    N = xy0.shape[1]
    dev = rgbs.device
    H, W = rgbs.shape[-2:]
    frame_maps = torch.cat([O.encoder(sd, 2 * (rgbs[0, t:t + 1] / 255.0) - 1.0, stride) for t in range(T)], dim=0)
    trajs_e = torch.zeros(1, T, N, 2, device=dev)
    trajs_e[0, 0] = xy0[0]
    cur = [0] * N
    hops = [[] for _ in range(N)]
    feat = [None] * N
    active = list(range(N))
    shape_only = rgbs[:, :1].expand(1, 8, 3, H, W)                                # forward reads rgbs for its SHAPE only when fmaps is given
    while active:
        n_act = len(active)
        idx = torch.tensor([[min(cur[n] + s, T - 1) for s in range(8)] for n in active], device=dev)   # :50-52 padding = last frame repeated
        fmaps = frame_maps[idx]                                                   # (n_act,8,128,H/s,W/s)
        start = torch.stack([trajs_e[0, cur[n], n] for n in active]).reshape(n_act, 1, 2)
        fi = None if feat[active[0]] is None else torch.stack([feat[n] for n in active]).reshape(n_act, 1, -1)
        preds, _, vis, ffeat = O.forward(sd, start, shape_only.expand(n_act, 8, 3, H, W), iters=iters, stride=stride,
                                         feat_init=fi, fmaps=fmaps)
        vis = torch.sigmoid(vis).cpu()                                            # (n_act,8,1)
        xys = preds[-1]                                                           # (n_act,8,1,2)
        nxt = []
        for j, n in enumerate(active):
            feat[n] = ffeat[j, 0]                                                 # :57 (the forward returns feat_init when given)
            S_local = min(8, T - cur[n])
            trajs_e[0, cur[n]:cur[n] + S_local, n] = xys[j, :S_local, 0]
            assert torch.isfinite(vis[j]).all()
            thr, si = 0.9, 7                                                      # :63-77
            while True:
                if vis[j, si, 0] > thr:
                    break
                si -= 1
                if si == 1:
                    thr -= 0.02
                    si = 7
            hops[n].append(si)
            cur[n] += si
            if cur[n] < T:
                nxt.append(n)
        active = nxt
    return trajs_e, hops
