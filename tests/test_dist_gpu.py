"""The multi-rank path on real devices: clips sharded over the ranks, ONE all-gather of [x, y, vis]; and the
secondary axis for B < G -- particles sharded on replicated or frame-sharded maps (``track_sharded_particles``,
``track_chained_sharded``).

* ``test_rccl_*`` need >= 2 visible GPUs and skip on the 1-GPU gpurun box; on a multi-GPU node they are the first thing
  that sends the real ``Pips`` through ``init_process_group("nccl", device_id=...)`` (RCCL over xGMI).
* ``test_rccl_one_rank`` runs on the 1-GPU box: one rank, backend nccl -- the communicator and every collective of pips_amd.dist
  execute in RCCL (as copies), bit-equal to the plain forward.
* ``test_two_ranks_share_one_gpu_gloo`` runs everywhere a GPU is visible: two ranks on cuda:0, gloo for the exchange (RCCL
  refuses two ranks on one device) -- the real model, the real shard / pack / gather / unpack code, only the transport differs.
  Its log goes to gpurun_out/ (copied to profiles/ by hand).
"""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, backend, share_device, q):
    import torch.distributed as dist
    from pips_amd import Pips, dist as pd
    from pips_amd.weights import init_state_dict
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dev = torch.device("cuda", 0 if share_device else rank)
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        m = Pips(stride=8)
        m.load_state_dict(init_state_dict(0, tamed=True))
        m = m.to(dev).eval()
        g = torch.Generator().manual_seed(21)
        B, N, H, W = 2 * world, 24, 128, 160
        rgbs = torch.randint(0, 256, (B, 8, 3, H, W), generator=g).float().to(dev)
        xys = (torch.rand(B, N, 2, generator=g) * torch.tensor([W - 1.0, H - 1.0])).to(dev)
        trajs, vis = pd.track_sharded(m, xys, rgbs, iters=3)
        ok = tuple(trajs.shape) == (B, 8, N, 2) and tuple(vis.shape) == (B, 8, N) and trajs.device == dev
        # what ONE rank computes for every shard, same call shapes: must be bit-identical to what came over the wire
        for r in range(world):
            lo, hi = pd.shard_range(B, r, world)
            out = m(xys[lo:hi], rgbs[lo:hi], iters=3)
            ok = ok and torch.equal(out[0][-1], trajs[lo:hi]) and torch.equal(out[2], vis[lo:hi])
        # ---- SURVEY 8(e) secondary axis (B < G): one clip, particles split over the ranks on replicated / frame-sharded maps
        from pips_amd import drivers
        xy1, clip = xys[:1, :23], rgbs[:1]                                  # 23 queries: padded to the world size inside
        xp, n = pd.pad_to_world(xy1, world, dim=1)
        cache = m.encode(clip)
        for mode in ("replicate", "frames") if 8 % world == 0 else ("replicate",):
            tp, vp = pd.track_sharded_particles(m, xy1, clip, iters=3, encode=mode)
            ok = ok and tuple(tp.shape) == (1, 8, 23, 2) and tuple(vp.shape) == (1, 8, 23)
            for r in range(world):
                lo, hi = pd.shard_range(xp.shape[1], r, world)
                out = m.track(cache, xp[:, lo:hi], iters=3)
                a, b = out[0][-1][:, :, :max(min(hi, n) - lo, 0)], tp[:, :, lo:min(hi, n)]
                # replicated maps: the shard's own single-rank result bit for bit; frame-sharded maps: encoded 8/G frames at
                # a time (other tile shapes than the 8-frame pass), per-frame InstanceNorm keeps that to round-off
                ok = ok and (torch.equal(a, b) if mode == "replicate" else float((a - b).abs().max()) < 1e-3)
        # ---- the chained long-video loop, particles split over the ranks (chain_demo.py:40: one particle at a time)
        video = torch.cat([clip, clip.flip(1)[:, :5]], dim=1)                   # T = 13
        xy0 = xy1[:, :5] * 0.8 + 10.0
        got = pd.track_chained_sharded(m, video, xy0, iters=3)
        xq, nq = pd.pad_to_world(xy0, world, dim=1)
        for r in range(world):
            lo, hi = pd.shard_range(xq.shape[1], r, world)
            ref = drivers.track_chained(m, video, xq[:, lo:hi], iters=3)
            ok = ok and torch.equal(ref[:, :, :max(min(hi, nq) - lo, 0)], got[:, :, lo:min(hi, nq)])
        ok = ok and tuple(got.shape) == (1, 13, 5, 2)
        q.put((rank, bool(ok), dist.get_backend()))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _run_ranks(world, backend, share_device):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, backend, share_device, q)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        res = [q.get(timeout=600) for _ in range(world)]
    finally:
        for p in procs:
            p.join(timeout=120)
            if p.is_alive():
                p.kill()
    return res


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs >= 2 GPUs (RCCL); the gpurun box has one")
def test_rccl_track_sharded_matches_single_rank():
    """N = all visible GPUs (<= 8) ranks, backend nccl (= RCCL): every rank ends with the full-batch result, bit-identical to
    what one rank computes shard by shard."""
    world = min(torch.cuda.device_count(), 8)
    res = _run_ranks(world, "nccl", share_device=False)
    assert sorted(r[0] for r in res) == list(range(world))
    assert all(r[1] for r in res), res
    assert all(r[2] == "nccl" for r in res)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs >= 2 GPUs (RCCL); the gpurun box has one")
def test_rccl_bench_line_two_ranks():
    """`python bench.py --gpus 2` as the driver launches it: the line must say the exchange went over RCCL."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.pop("PIPS_BENCH_BACKEND", None)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                          "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2",
                          "--steps", "3", "--warmup", "1"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    res = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert res["n_gpus"] == 2 and res["scaling"] == "weak" and res["value"] > 0
    assert res["config"]["collective"].endswith("nccl (RCCL)")
    assert res["collective_ms"]["median_ms"] > 0


def test_rccl_one_rank():
    """What ONE GPU allows of the RCCL path (round 6): ``init_process_group("nccl", world_size=1, device_id=cuda:0)`` loads
    librccl, creates the communicator and runs every collective of pips_amd.dist -- the packed ``all_gather_into_tensor`` of
    ``track_sharded`` (clips), ``track_sharded_particles`` on replicated AND frame-sharded maps (one all-gather per pyramid
    level + one on the particle axis) and ``track_chained_sharded`` -- on device tensors through the real ``nccl`` backend
    (pips_amd.dist runs its collectives whenever a process group exists, also with one rank).  Bit-equal to the plain forward.
    It cannot show scaling; it shows that the RCCL branch executes."""
    res = _run_ranks(1, "nccl", share_device=False)
    assert [r[0] for r in res] == [0] and res[0][1], res
    assert res[0][2] == "nccl"
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "rccl_one_rank.log"), "w") as f:
        f.write("tests/test_dist_gpu.py::test_rccl_one_rank: init_process_group('nccl', world_size=1, device_id=cuda:0); track_sharded, "
                "track_sharded_particles (replicate, frames), track_chained_sharded: every collective through RCCL on device tensors\n")
        f.write(f"rank 0: gathered == plain single-process results bit for bit: {res[0][1]}  backend {res[0][2]}\n")


def test_two_ranks_share_one_gpu_gloo():
    """Two ranks on cuda:0 with the REAL model; the all-gather goes through gloo's host path."""
    res = _run_ranks(2, "gloo", share_device=True)
    assert sorted(r[0] for r in res) == [0, 1] and all(r[1] for r in res), res
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "two_ranks_one_gpu_gloo.log"), "w") as f:
        f.write("tests/test_dist_gpu.py::test_two_ranks_share_one_gpu_gloo: 2 ranks on cuda:0, real Pips, track_sharded (clips), "
                "track_sharded_particles (replicated and frame-sharded maps) and track_chained_sharded over gloo\n")
        for r in sorted(res):
            f.write(f"rank {r[0]}: gathered == per-shard single-rank results bit for bit: {r[1]}  backend {r[2]}\n")
