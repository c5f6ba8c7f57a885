"""Multi-GPU: one process per GPU.

The reference has no distributed inference path (only ``nn.DataParallel`` for training,
train.py:254).  Two axes shard without any exchange inside the forward:

* **clips** (primary, ``track_sharded``): clips are fully independent in ``Pips.forward`` --
  InstanceNorm is per frame, correlation and mixer are per (clip, particle) -- so each rank runs
  whole clips and the only exchange is one all-gather of the final ``[x, y, vis_logit]`` per
  (clip, frame, particle): 196 KB per rank at B/G=8, N=256.
* **particles** (secondary, for B < G: ``track_sharded_particles``, ``track_chained_sharded``):
  given the feature maps, particles are independent (the reference's own callers loop over them:
  chain_demo.py:40, test_on_davis.py:116-118).  Every rank holds the maps of the whole clip --
  encoded redundantly (``encode="replicate"``: 2.8 ms at BASELINE configs[1], no exchange) or each rank
  encoding T/G frames followed by one all-gather per pyramid level (``encode="frames"``: exact, because
  InstanceNorm is per frame; 11.7 MB at configs[1]) -- tracks N/G particles on them and the final
  ``[x, y, vis_logit]`` are gathered on the particle axis.

With the ``nccl`` backend the collectives are RCCL over xGMI; all of them are latency-bound
(one direct write per peer), none sits inside the update loop.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def _world(group=None):
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def _live(group=None) -> bool:
    """A process group exists: the collectives RUN, also with one rank (a one-rank ``all_gather_into_tensor`` is a copy through
    the backend -- that is how the 1-GPU box executes the RCCL branch, tests/test_dist_gpu.py::test_rccl_one_rank).  Without a
    process group the sharded entry points are plain single-process calls."""
    return dist.is_available() and dist.is_initialized()


def shard_range(total: int, rank: int, world: int):
    """Contiguous range of ``rank`` out of ``total`` clips / particles; requires an even split so the packed
    all-gather needs no padding (``pad_to_world`` pads a batch that does not divide)."""
    if total % world != 0:
        raise ValueError(f"{total} is not divisible by the number of ranks {world}; pad with pips_amd.dist.pad_to_world")
    per = total // world
    return rank * per, (rank + 1) * per


def pad_to_world(x: torch.Tensor, world: int, dim: int = 0):
    """Pad ``x`` along ``dim`` to the next multiple of ``world`` by repeating its last slice (a valid clip / query, so the
    padded work is well defined).  Returns ``(padded, original_length)``; cut the gathered result back with
    ``out.narrow(dim, 0, original_length)``."""
    n = x.shape[dim]
    if n == 0:
        raise ValueError("pad_to_world: empty axis")
    pad = (-n) % world
    if pad == 0:
        return x, n
    last = x.narrow(dim, n - 1, 1)
    return torch.cat([x, last.expand(*[pad if d == dim % x.dim() else -1 for d in range(x.dim())])], dim=dim).contiguous(), n


def pack_result(trajs_e: torch.Tensor, vis_e: torch.Tensor) -> torch.Tensor:
    """(b,S,N,2), (b,S,N) -> (b,S,N,3) contiguous."""
    return torch.cat([trajs_e, vis_e.unsqueeze(-1)], dim=-1).contiguous()


def _all_gather_cat(mine: torch.Tensor, dim: int, group=None) -> torch.Tensor:
    """ONE ``all_gather_into_tensor`` of equally shaped contiguous tensors, concatenated along ``dim``.  Both backends in
    use (nccl = RCCL, gloo) implement it; an error of the collective is an error of the job and is not retried on another
    collective."""
    world = dist.get_world_size(group)
    mine = mine.contiguous()
    if mine.is_cuda and dist.get_backend(group) == "gloo":
        # gloo moves device tensors through the host anyway: do it explicitly (tests with two ranks on one GPU)
        host = torch.empty((world * mine.shape[0],) + tuple(mine.shape[1:]), dtype=mine.dtype)
        dist.all_gather_into_tensor(host, mine.cpu(), group=group)
        out = host.to(mine.device)
    else:
        out = torch.empty((world * mine.shape[0],) + tuple(mine.shape[1:]), dtype=mine.dtype, device=mine.device)
        dist.all_gather_into_tensor(out, mine, group=group)
    if dim == 0:
        return out
    # (world, ..., n, ...) -> (..., world * n, ...): rank-major along ``dim``
    return torch.cat(list(out.view(world, *mine.shape).unbind(0)), dim=dim)


def all_gather_result(trajs_e: torch.Tensor, vis_e: torch.Tensor, group=None, dim: int = 0):
    """Every rank receives the full ``trajs_e (B,S,N,2)`` and ``vis_e (B,S,N)`` from per-rank shards along ``dim``
    (0 = clips, 2 = particles).  One ``all_gather_into_tensor`` of the packed ``[x, y, vis]``."""
    if not _live(group):
        return trajs_e, vis_e
    out = _all_gather_cat(pack_result(trajs_e, vis_e), dim, group)
    return out[..., :2].contiguous(), out[..., 2].contiguous()


def track_sharded(model, xys, rgbs, iters=6, group=None, **kw):
    """Run ``model`` on this rank's slice of the batch and gather the final trajectories.
    ``xys`` (B,N,2) / ``rgbs`` (B,S,3,H,W) hold the FULL batch on every rank (or at least
    this rank's slice must be valid)."""
    rank, world = _world(group)
    lo, hi = shard_range(xys.shape[0], rank, world)
    out = model(xys[lo:hi], rgbs[lo:hi], iters=iters, **kw)
    return all_gather_result(out[0][-1], out[2], group=group)


def encode_sharded(model, rgbs, group=None):
    """``model.encode(rgbs (B,T,3,H,W))`` with the T frames of every clip split over the ranks: each rank encodes T/G
    frames, then one all-gather per pyramid level rebuilds the whole cache on every rank.  Exact: InstanceNorm is per frame
    (nets/pips.py:153-157), so a frame's maps do not depend on which frames share its pass.  T must divide by the world
    size (``pad_to_world(rgbs, world, dim=1)`` otherwise)."""
    from . import _lib, ops
    from .pips import FeatureCache
    rank, world = _world(group)
    if not _live(group):
        return model.encode(rgbs)
    B, T, _, H, W = rgbs.shape
    lo, hi = shard_range(T, rank, world)
    part = model.encode(rgbs[:, lo:hi])                                   # (B, T/G) frames, frame index b * T/G + t
    lib, st, per = _lib.load(), int(model.stride), hi - lo
    pyr = torch.empty(lib.pips_pyramid_floats(B * T, H, W, st), dtype=torch.float32, device=part.pyr.device)
    for dst, src in zip(ops.pyramid_levels(pyr, B * T, H, W, st), ops.pyramid_levels(part.pyr, B * per, H, W, st)):
        # (B*per, h, w, C) per rank -> (world, B, per, ...) -> clip-major (B, world*per = T, ...)
        g = _all_gather_cat(src.reshape(B, per, *src.shape[1:]), 1, group)
        dst.copy_(g.reshape(B * T, *src.shape[1:]))
    if part.bf16_maps:                                                    # the bf16 mirror follows the whole buffer's layout
        import ctypes as C
        with torch.cuda.device(pyr.device):
            _lib.check(lib.pips_pyramid_mirror(_lib.ptr(pyr), B * T, H, W, st,
                                               C.c_void_p(torch.cuda.current_stream().cuda_stream)), "pips_pyramid_mirror")
    return FeatureCache(pyr, B, T, H, W, st, bf16_maps=part.bf16_maps)


def track_sharded_particles(model, xys, rgbs, iters=6, group=None, encode="replicate", **kw):
    """The secondary axis of SURVEY §8e (B < number of GPUs: BASELINE configs[1] B=1, configs[3] B=4): every rank holds the
    clip's maps (``encode="replicate"``: each rank encodes all frames; ``"frames"``: ``encode_sharded``), tracks its N/G
    particles on them with ``model.track`` and the final ``[x, y, vis_logit]`` are gathered on the particle axis.
    ``xys (B,N,2)``, ``rgbs (B,S,3,H,W)`` are the full inputs on every rank; N is padded to the world size internally.
    Returns ``(trajs_e (B,S,N,2), vis_e (B,S,N))`` on every rank."""
    rank, world = _world(group)
    if encode not in ("replicate", "frames"):
        raise ValueError("encode must be 'replicate' or 'frames'")
    cache = encode_sharded(model, rgbs, group) if encode == "frames" else model.encode(rgbs)
    xp, n = pad_to_world(xys, world, dim=1)
    lo, hi = shard_range(xp.shape[1], rank, world)
    out = model.track(cache, xp[:, lo:hi], iters=iters, **kw)
    trajs, vis = all_gather_result(out[0][-1], out[2], group=group, dim=2)
    return trajs[:, :, :n].contiguous(), vis[:, :, :n].contiguous()


def track_chained_sharded(model, rgbs, xy0, iters=6, group=None):
    """``drivers.track_chained`` (chain_demo.py:40-83) with the particles split over the ranks -- the reference's loop
    handles one particle at a time (chain_demo.py:40), so their chains are independent: every rank encodes the video
    (per-frame maps, no exchange), chains N/G particles and one all-gather on the particle axis collects
    ``trajs_e (1,T,N,2)``."""
    from . import drivers
    rank, world = _world(group)
    xp, n = pad_to_world(xy0, world, dim=1)
    lo, hi = shard_range(xp.shape[1], rank, world)
    mine = drivers.track_chained(model, rgbs, xp[:, lo:hi], iters=iters)          # (1,T,n/G,2)
    if not _live(group):
        return mine
    return _all_gather_cat(mine, 2, group)[:, :, :n].contiguous()
