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


def test_bench_rejects_world_size_mismatch():
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", PIPS_BENCH_FAKE="1")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode != 0 and "WORLD_SIZE=1" in (out.stderr + out.stdout)
