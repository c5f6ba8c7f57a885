"""GPU: the encoder/tracker split (Pips.encode / Pips.track, C entry pips_track) and the two
caller loops rebuilt on it (dense grid: test_on_davis.py:103-130; chaining: chain_demo.py:40-83)."""
import pytest
import torch

import cases as G

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _model(sd, stride=8):
    from pips_amd import Pips
    m = Pips(stride=stride)
    m.load_state_dict(sd)
    return m.to(DEV).eval()


def test_track_on_cache_equals_forward(weights_raw):
    m = _model(weights_raw)
    xys, rgbs, _, _ = G.make_inputs(dict(B=2, N=33, H=128, W=160))
    ref = m(xys.to(DEV), rgbs.to(DEV), iters=3, return_feat=True)
    cache = m.encode(rgbs.to(DEV))
    got = m.track(cache, xys.to(DEV), iters=3, return_feat=True)
    for a, b in zip(ref[0] + ref[1] + [ref[2], ref[3]], got[0] + got[1] + [got[2], got[3]]):
        assert torch.equal(a, b)                       # same kernels, same order: bit-identical


def test_encode_in_passes_matches_single_pass(weights_raw):
    m = _model(weights_raw)
    _, rgbs, _, _ = G.make_inputs(dict(B=1, N=1, H=128, W=160))
    video = torch.cat([rgbs, rgbs.flip(1), rgbs[:, :4]], dim=1).to(DEV)          # T = 20
    from pips_amd import _lib
    n = _lib.load().pips_pyramid_mirror_offset(20, 128, 160, 8)      # the fp32 levels (the bf16 mirror behind them is written in the bf16 mode only)
    a = m.encode(video, frames_per_pass=64).pyr[:n]
    b = m.encode(video, frames_per_pass=8).pyr[:n]
    assert float((a - b).abs().max()) < 1e-4            # per-frame InstanceNorm: only tile-order noise
    # bf16 mode: the mirror of a cache encoded in passes equals the mirror of the single pass (rewritten for the whole buffer)
    m.encoder_dtype = m.mixer_dtype = torch.bfloat16
    ca, cb = m.encode(video, frames_per_pass=64), m.encode(video, frames_per_pass=8)
    assert ca.bf16_maps and cb.bf16_maps
    for c in (ca, cb):                                               # the mirror IS bf16(fp32 levels), bit for bit
        assert torch.equal(c.pyr[n:n + n // 2].view(torch.bfloat16), c.pyr[:n].bfloat16())
    rel = float((ca.pyr[:n] - cb.pyr[:n]).pow(2).mean().sqrt() / cb.pyr[:n].pow(2).mean().sqrt())
    assert rel < 2e-2                                                # different tile shapes at 20 / 8 frames: bf16 rounding noise
    xys = torch.tensor([[[40.0, 50.0], [100.0, 64.0]]], device=DEV)
    ta = m.track(ca, xys, iters=2, win_start=torch.tensor([[3, 9]]))
    tb = m.track(cb, xys, iters=2, win_start=torch.tensor([[3, 9]]))
    assert float((ta[0][-1] - tb[0][-1]).abs().max()) < 0.5          # raw weights, bf16 mode: same track up to bf16 noise


def test_dense_chunks_match_single_call(weights_tamed):
    from pips_amd import drivers
    m = _model(weights_tamed)
    _, rgbs, _, _ = G.make_inputs(dict(B=1, N=1, H=128, W=160))
    gy, gx = torch.meshgrid(torch.arange(4, 128, 8.0), torch.arange(4, 160, 8.0), indexing="ij")
    xys = torch.stack([gx.reshape(-1), gy.reshape(-1)], -1).unsqueeze(0)          # 320 points (davis.py:103-105)
    t_all, v_all = drivers.track_dense(m, rgbs.to(DEV), xys.to(DEV), iters=6)
    t_chk, v_chk = drivers.track_dense(m, rgbs.to(DEV), xys.to(DEV), iters=6, chunk=96)
    full = m(xys.to(DEV), rgbs.to(DEV), iters=6)
    assert float((t_all - full[0][-1]).abs().max()) == 0.0
    assert float((t_chk - t_all).abs().max()) < 1e-3 and float((v_chk - v_all).abs().max()) < 1e-3


def test_chained_tracking_matches_reference_loop(weights_tamed):
    from pips_amd import drivers
    from oracle import chain_oracle
    g = torch.Generator().manual_seed(4)
    T, H, W, N = 21, 128, 160, 5
    base = torch.randint(0, 256, (1, 1, 3, H, W), generator=g).float()
    # a slowly changing video so visibility logits vary between frames
    video = torch.cat([(base * (1 - 0.03 * t) + 7.0 * t).clamp(0, 255).round() for t in range(T)], dim=1)
    video = (video + torch.randint(0, 40, video.shape, generator=g).float()).clamp(0, 255)
    xy0 = torch.rand(1, N, 2, generator=g) * torch.tensor([W - 17.0, H - 17.0]) + 8.0
    ref, hops = chain_oracle.chain(weights_tamed, video, xy0, iters=6, stride=8)
    got = drivers.track_chained(_model(weights_tamed), video.to(DEV), xy0.to(DEV), iters=6).cpu()
    print("hop sequences:", hops)
    assert any(len(h) > 2 for h in hops)
    assert tuple(got.shape) == (1, T, N, 2)
    err = float((got - ref).abs().max())
    print("chained max |dtraj| px:", err)
    assert err < 1e-3


def test_chained_tracking_against_reference_loop_text_golden(weights_tamed):
    """drivers.track_chained held DIRECTLY to tests/golden/chain_t13.npz -- the output of the reference's own loop text
    (chain_demo.py:39-83 executed verbatim with the unmodified model, tests/golden/make_chain_golden.py): same window
    starts, same hop steps, trajectories within 1e-3 px.  (Until round 5 this was transitive: HIP = oracle = golden.)"""
    import os
    import numpy as np
    from pips_amd import drivers
    case = G.CHAIN_CASE
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(G.__file__)), "chain_t13.npz"))
    video, xy0 = G.make_chain_inputs(case)
    got, hops = drivers.track_chained(_model(weights_tamed, case["stride"]), video.to(DEV), xy0.to(DEV), iters=case["iters"],
                                      return_hops=True)
    starts = []
    for seq in hops:                                  # the generator logged each window's first frame, particle by particle
        cur = 0
        for si in seq:
            starts.append(cur)
            cur += si
    assert starts == gold["window_starts"].tolist()
    assert [si for seq in hops for si in seq[:-1]] == gold["hop_steps"].tolist()
    assert [len(seq) - 1 for seq in hops] == gold["hops_per_particle"].tolist()
    err = float((got.cpu() - torch.from_numpy(gold["trajs_e"])).abs().max())
    print("HIP chained tracking vs reference loop text: max |dtraj| = %.2e px" % err)
    assert tuple(got.shape) == (1, case["T"], case["N"], 2) and err < 1e-3


def test_skip_scan_matches_reference_scan():
    from pips_amd import drivers
    g = torch.Generator().manual_seed(0)
    vis = torch.rand(8, 500, generator=g)
    vis[:, :50] *= 0.3                                   # force several threshold decrements
    got = drivers.skip_scan(vis.to(DEV)).cpu()
    for i in range(vis.shape[1]):
        thr, si = 0.9, 7
        while True:
            if vis[si, i] > thr:
                break
            si -= 1
            if si == 1:
                thr -= 0.02
                si = 7
        assert int(got[i]) == si
