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
def chain(sd, rgbs, xy0, iters=6, stride=8):
    """rgbs (1,T,3,H,W), xy0 (1,N,2) -> trajs_e (1,T,N,2), list of hop sequences per particle."""
    B, T = rgbs.shape[:2]
    N = xy0.shape[1]
    trajs_e = torch.zeros(B, T, N, 2)
    hops = []
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
            preds, _, vis, ffeat = O.forward(sd, traj_e[:, cur].reshape(1, -1, 2), rgb_seq, iters=iters,
                                             stride=stride, feat_init=feat_init)
            feat_init = ffeat                                             # :57
            vis = torch.sigmoid(vis)
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
