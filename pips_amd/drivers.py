"""Caller-side loops of the reference re-built on cached maps.

* ``track_dense``   -- test_on_davis.py:103-130: many query points on one clip.  The reference
  re-runs the encoder for every chunk of 256 points; here the clip is encoded once.
* ``track_chained`` -- chain_demo.py:40-83 / test_on_badja.py:64-112: visibility-aware
  chaining of 8-frame windows over a long video.  The reference re-encodes 8 frames per particle
  and per hop; here every video frame is encoded once and all live particles advance together,
  each with its own window start.

Host logic only (a few tiny torch ops on (8,N) tensors); all model arithmetic is in
libpips_hip.so through ``Pips.encode`` / ``Pips.track``.
"""
from __future__ import annotations

import torch


@torch.no_grad()
def track_dense(model, rgbs, xys, iters=6, chunk=None):
    """rgbs (B,8,3,H,W), xys (B,N,2) -> (trajs_e (B,8,N,2), vis_e (B,8,N) logits)."""
    cache = model.encode(rgbs)
    N = xys.shape[1]
    chunk = N if chunk is None else chunk
    trajs, viss = [], []
    for n0 in range(0, N, chunk):
        preds, _, vis, _ = model.track(cache, xys[:, n0:n0 + chunk], iters=iters)
        trajs.append(preds[-1])
        viss.append(vis)
    return torch.cat(trajs, dim=2), torch.cat(viss, dim=2)


def _threshold_table(n=64):
    """thr after k decrements exactly as chain_demo.py:64,75 computes it (python doubles),
    rounded to float32 as the tensor comparison ``vis[0,si] > thr`` does."""
    thr, out = 0.9, []
    for _ in range(n):
        out.append(thr)
        thr -= 0.02
    return torch.tensor(out, dtype=torch.float64).to(torch.float32)


def skip_scan(vis):
    """vis (8,n) sigmoid confidences -> si (n,) int64: chain_demo.py:63-77.  Frames 7..2 are
    tested against thr (0.9, lowered by 0.02 whenever the scan reaches frame 1); the LATEST
    frame above the threshold wins."""
    thr = _threshold_table().to(vis.device)
    cand = vis[2:8].unsqueeze(0) > thr.view(-1, 1, 1)                 # (K,6,n)
    anyk = cand.any(dim=1)                                            # (K,n)
    kfirst = torch.argmax(anyk.to(torch.int32), dim=0)                # first threshold that admits a frame
    n = vis.shape[1]
    sel = cand[kfirst, :, torch.arange(n, device=vis.device)]         # (n,6)
    last = 5 - torch.argmax(sel.flip(1).to(torch.int32), dim=1)       # latest admitted frame
    return last + 2


@torch.no_grad()
def track_chained(model, rgbs, xy0, iters=6, return_hops=False):
    """rgbs (1,T,3,H,W), xy0 (1,N,2) px at frame 0 -> trajs_e (1,T,N,2) (chain_demo.run_model).
    ``return_hops=True``: also the list, per particle, of the frame steps ``si`` its windows advanced by
    (chain_demo.py:63-79) -- what a parity test compares hop for hop."""
    assert rgbs.shape[0] == 1, "the reference chains one video at a time (chain_demo.py:24)"
    assert model.S == 8, "chain_demo.py's visibility scan (frames 7..2 of an 8-frame window) is written for S = 8"
    dev = rgbs.device
    T, N, S = rgbs.shape[1], xy0.shape[1], 8
    cache = model.encode(rgbs)
    # S - 1 frames of padding behind the video: a window that runs past the end is written whole and cut off on return
    # (no per-row masks, no host round trips inside a hop)
    trajs = torch.zeros(1, T + S - 1, N, 2, dtype=torch.float32, device=dev)
    trajs[0, 0] = xy0[0].to(dev)
    offs = torch.arange(S, device=dev).unsqueeze(1)                                # (S,1)
    cur = torch.zeros(N, dtype=torch.int64, device=dev)
    active = torch.arange(N, device=dev)
    feat = None
    log = []
    while active.numel() > 0:
        c = cur[active]
        start_xy = trajs[0, c, active].unsqueeze(0)                               # traj_e[:,cur_frame]
        fi = None if feat is None else feat[active].unsqueeze(0)
        preds, _, vis, ffeat, _ = model.track(cache, start_xy, iters=iters, feat_init=fi,
                                              win_start=c.to(torch.int32).unsqueeze(0), return_feat=True)
        if feat is None:
            feat = ffeat[0].clone()                                              # carried forever (:57)
        xys = preds[-1][0]                                                        # (8,n,2)
        trajs[0, c.unsqueeze(0) + offs, active.unsqueeze(0).expand(S, -1)] = xys   # traj_e[cur:cur+8] = xys[:S_local]
        si = skip_scan(torch.sigmoid(vis[0]))
        cur[active] = c + si
        if return_hops:
            log.append((active, si))
        active = active[cur[active] < T]                                          # (one host sync per hop: the live count)
    out = trajs[:, :T].contiguous()
    if not return_hops:
        return out
    hops = [[] for _ in range(N)]
    for act, si in log:
        for n, k in zip(act.tolist(), si.tolist()):
            hops[n].append(k)
    return out, hops
