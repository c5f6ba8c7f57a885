"""Multi-GPU: clips shard on the batch axis, one process per GPU.

The reference has no distributed inference path (only ``nn.DataParallel`` for training,
train.py:254).  Clips are fully independent in ``Pips.forward`` -- InstanceNorm is per
frame, correlation and mixer are per (clip, particle) -- so each rank runs whole clips
and the only exchange is one all-gather of the final ``[x, y, vis_logit]`` per
(clip, frame, particle): 196 KB per rank at B/G=8, N=256.  With the ``nccl`` backend
this is RCCL over xGMI; latency-bound, one direct write per peer.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_range(total_clips: int, rank: int, world: int):
    """Contiguous clip range of ``rank``; requires an even split so the packed all-gather
    needs no padding (callers pad the batch if necessary)."""
    if total_clips % world != 0:
        raise ValueError(f"B={total_clips} is not divisible by the number of ranks {world}")
    per = total_clips // world
    return rank * per, (rank + 1) * per


def pack_result(trajs_e: torch.Tensor, vis_e: torch.Tensor) -> torch.Tensor:
    """(b,S,N,2), (b,S,N) -> (b,S,N,3) contiguous."""
    return torch.cat([trajs_e, vis_e.unsqueeze(-1)], dim=-1).contiguous()


def all_gather_result(trajs_e: torch.Tensor, vis_e: torch.Tensor, group=None):
    """Every rank receives the full-batch ``trajs_e (B,S,N,2)`` and ``vis_e (B,S,N)``.
    One ``all_gather_into_tensor`` (falls back to ``all_gather`` on backends without it)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return trajs_e, vis_e
    world = dist.get_world_size(group)
    mine = pack_result(trajs_e, vis_e)
    out = torch.empty((world * mine.shape[0],) + tuple(mine.shape[1:]), dtype=mine.dtype, device=mine.device)
    try:
        dist.all_gather_into_tensor(out, mine, group=group)
    except (RuntimeError, NotImplementedError):
        parts = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine, group=group)
        out = torch.cat(parts, dim=0)
    return out[..., :2].contiguous(), out[..., 2].contiguous()


def track_sharded(model, xys, rgbs, iters=6, group=None, **kw):
    """Run ``model`` on this rank's slice of the batch and gather the final trajectories.
    ``xys`` (B,N,2) / ``rgbs`` (B,S,3,H,W) hold the FULL batch on every rank (or at least
    this rank's slice must be valid)."""
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    lo, hi = shard_range(xys.shape[0], rank, world)
    out = model(xys[lo:hi], rgbs[lo:hi], iters=iters, **kw)
    return all_gather_result(out[0][-1], out[2], group=group)
