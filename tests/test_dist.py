"""CPU, world_size 2, gloo: the batch-shard + packed all-gather path of pips_amd.dist."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pips_amd import dist as pd


class _FakeTracker:
    """Stands in for Pips on CPU: deterministic function of the clip content."""

    def __call__(self, xys, rgbs, iters=6, **kw):
        B, N, _ = xys.shape
        S = rgbs.shape[1]
        base = xys.reshape(B, 1, N, 2).repeat(1, S, 1, 1) + rgbs.mean(dim=(2, 3, 4)).reshape(B, S, 1, 1)
        preds = [base + i for i in range(iters)]
        vis = base.sum(-1)
        return preds, [base, base] + preds + [preds[-1]] * 2, vis, None


class _FakeCache:
    def __init__(self, rgbs):
        self.m = rgbs.float().mean(dim=(2, 3, 4))                      # (B,T) per-frame content
        self.B, self.T = self.m.shape


class _FakeModel(_FakeTracker):
    """encode / track stand-in with the real signatures: a particle's result depends on its own query, its window start and
    the frames of its window only -- the independence the particle-axis sharding relies on."""
    S = 8

    def encode(self, rgbs):
        return _FakeCache(rgbs)

    def track(self, cache, xys, coords_init=None, feat_init=None, iters=3, win_start=None, return_feat=False):
        B, N, _ = xys.shape
        ws = torch.zeros(B, N, dtype=torch.long) if win_start is None else win_start.long()
        t = (ws.unsqueeze(1) + torch.arange(8).view(1, 8, 1)).clamp(max=cache.T - 1)               # (B,8,N)
        fm = torch.gather(cache.m.unsqueeze(2).expand(B, cache.T, N), 1, t)                      # (B,8,N)
        base = xys.reshape(B, 1, N, 2) + 0.01 * fm.unsqueeze(-1) * torch.arange(8).view(1, 8, 1, 1)
        preds = [base + 0.1 * i for i in range(iters)]
        vis = torch.sin(base.sum(-1) * 3.0) * 4.0                                                  # logits of both signs
        out = (preds, [base, base] + preds + [preds[-1]] * 2, vis)
        ff = xys.new_zeros(B, N, 128) if feat_init is None else feat_init
        return out + ((ff, None) if return_feat else (None,))


def _worker_particles(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pips_amd import drivers
    g = torch.Generator().manual_seed(3)
    m = _FakeModel()
    ok = True
    for N in (6, 7):                                                   # 7: padded to 8, cut back after the gather
        xys = torch.rand(2, N, 2, generator=g) * 50
        rgbs = torch.rand(2, 8, 3, 16, 16, generator=g)
        trajs, vis = pd.track_sharded_particles(m, xys, rgbs, iters=3)
        full = m.track(m.encode(rgbs), xys, iters=3)
        ok = ok and torch.equal(trajs, full[0][-1]) and torch.equal(vis, full[2]) and tuple(trajs.shape) == (2, 8, N, 2)
        video = torch.rand(1, 21, 3, 16, 16, generator=g)
        xy0 = torch.rand(1, N, 2, generator=g) * 50
        got = pd.track_chained_sharded(m, video, xy0, iters=2)
        ok = ok and torch.equal(got, drivers.track_chained(m, video, xy0, iters=2)) and tuple(got.shape) == (1, 21, N, 2)
    # the gather helper along an inner axis: rank-major concatenation
    mine = torch.full((2, 3, 4), float(rank))
    cat = pd._all_gather_cat(mine, 1)
    ok = ok and tuple(cat.shape) == (2, 3 * world, 4) and all(float(cat[0, 3 * r, 0]) == r for r in range(world))
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def _spawn(fn, world=2):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=fn, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    return res


def test_particle_axis_sharding_world2():
    """SURVEY 8(e) secondary axis: N/G particles per rank on replicated maps, one gather on the particle axis -- for the
    one-shot tracker and for the chained long-video loop (chain_demo.py:40 iterates particles independently)."""
    assert _spawn(_worker_particles) == {0: True, 1: True}


def test_pad_to_world():
    x = torch.arange(10.0).reshape(1, 5, 2)
    p, n = pd.pad_to_world(x, 4, dim=1)
    assert n == 5 and tuple(p.shape) == (1, 8, 2) and torch.equal(p[:, :5], x) and torch.equal(p[:, 5:], x[:, 4:5].expand(1, 3, 2))
    q, n = pd.pad_to_world(x, 5, dim=1)
    assert q is x and n == 5
    assert pd.pad_to_world(x, 3, dim=0)[0].shape[0] == 3


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(0)
    xys = torch.rand(4, 6, 2, generator=g)
    rgbs = torch.rand(4, 8, 3, 16, 16, generator=g)
    trajs, vis = pd.track_sharded(_FakeTracker(), xys, rgbs, iters=3)
    full = _FakeTracker()(xys, rgbs, iters=3)
    ok = torch.equal(trajs, full[0][-1]) and torch.equal(vis, full[2]) and tuple(trajs.shape) == (4, 8, 6, 2)
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_all_gather_world2():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
    assert res == {0: True, 1: True}


def test_shard_range():
    assert pd.shard_range(64, 3, 8) == (24, 32)
    try:
        pd.shard_range(7, 0, 2)
        assert False
    except ValueError:
        pass


def test_bench_spawns_its_own_ranks():
    """`python bench.py --gpus 2` with WORLD_SIZE unset re-executes itself under torch.distributed.run with two
    ranks and reports n_gpus = 2 (CPU stand-in model + gloo: the launcher, barrier, max-over-ranks and all-gather
    plumbing of the real run)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(PIPS_BENCH_FAKE="1", PIPS_BENCH_BACKEND="gloo")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    res = json.loads(line)
    assert res["n_gpus"] == 2 and res["steps"] == 2 and res["value"] > 0 and res["scaling"] == "weak"
    assert res["ms_per_step_median"] > 0
    # a multi-rank line carries the config it is about: the exchange alone, and BASELINE
    # configs[2] weak (8 clips per GPU) and strong (64 clips in total) -- here through the CPU stand-in
    assert res["collective_ms"]["median_ms"] > 0 and res["collective_ms"]["reps"] == 20
    c3 = res["config3"]
    assert c3["n_gpus"] == 2 and c3["weak"]["clips_per_gpu"] == 8 and c3["weak"]["clips_total"] == 16 and c3["weak"]["value"] > 0
    assert c3["strong"]["clips_total"] == 64 and c3["strong"]["clips_per_gpu"] == 32 and c3["strong"]["value"] > 0


def test_bench_config4_particle_sharded_line():
    """`python bench.py --gpus 2 --config 4`: BASELINE configs[3] has B = 4 < 8 GPUs, so the line shards the PARTICLES
    (pips_amd.dist.track_sharded_particles) -- strong scaling, the job's updates do not grow with the ranks."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(PIPS_BENCH_FAKE="1", PIPS_BENCH_BACKEND="gloo")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--config", "4", "--steps", "2", "--warmup",
                          "1"], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    res = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert res["n_gpus"] == 2 and res["scaling"] == "strong" and res["value"] > 0
    assert "particle-sharded x2" in res["config"]["parallelism"] and "particle axis" in res["config"]["collective"]
    assert abs(res["value"] - 4 * 8 * 64 * 6 * 2 / (res["ms_per_step"] * 2 / 1e3)) < 1e-6 * res["value"]


def test_bench_rejects_world_size_mismatch():
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", PIPS_BENCH_FAKE="1")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode != 0 and "WORLD_SIZE=1" in (out.stderr + out.stdout)
