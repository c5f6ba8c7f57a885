"""GPU parity at the geometries of BASELINE configs[3] (720x1280, N=4096 on a 64x64 grid: the dense-query path
through the LDS-tiled gather) and configs[4] (360x640, stride 4, chained windows over a longer video)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _grid(n, h, w, margin=8.0):
    k = int(round(n ** 0.5))
    gy, gx = torch.meshgrid(torch.linspace(margin, h - margin, k), torch.linspace(margin, w - margin, k), indexing="ij")
    return torch.stack([gx.reshape(-1), gy.reshape(-1)], -1)


def _pm(t):
    B, S, N, X = t.shape
    return t.permute(0, 2, 1, 3).reshape(B * N * S, X).contiguous()


def test_config4_gather_direct_and_tiled_vs_oracle():
    """CorrBlock.corr + sample at 90x160 maps, N=4096 grid (+ per-frame jitter, some points pushed outside):
    both the direct kernel and the LDS-tiled kernels against the oracle, all 4096 x 8 x 196 taps."""
    from pips_amd import ops, _lib
    from oracle import pips_oracle as O
    lib = _lib.load()
    B, H8, W8, N = 1, 90, 160, 4096
    g = torch.Generator().manual_seed(21)
    fmaps = torch.randn(B, 8, 128, H8, W8, generator=g)
    ffeats = torch.randn(B, 8, N, 128, generator=g)
    coords = (_grid(N, H8 * 8, W8 * 8) / 8.0).reshape(1, 1, N, 2).repeat(B, 8, 1, 1) + torch.randn(B, 8, N, 2, generator=g) * 1.5
    coords[0, :, 0] = torch.tensor([-6.0, 3.0])                      # window half outside / fully outside the map
    coords[0, :, 1] = torch.tensor([W8 + 40.0, H8 + 2.0])
    coords[0, :, 2] = torch.tensor([31.999998, 16.0])                # on tile / pixel boundaries
    coords[0, :, 3] = torch.tensor([16.0, 47.999996])
    pyr_ref = O.build_pyramid(fmaps)
    ref = torch.cat([O.corr_sample(pyr_ref, ffeats[:, :, n0:n0 + 512], coords[:, :, n0:n0 + 512])
                     for n0 in range(0, N, 512)], dim=2)             # (B,8,N,196), chunked: 236 MB of volume at a time
    buf = torch.zeros(lib.pips_pyramid_floats(B * 8, H8 * 8, W8 * 8, 8))
    for l, p in enumerate(pyr_ref):
        off = lib.pips_pyramid_offset(B * 8, H8 * 8, W8 * 8, 8, l)
        flat = p.reshape(B * 8, 128, p.shape[-2], p.shape[-1]).permute(0, 2, 3, 1).reshape(-1)
        buf[off:off + flat.numel()] = flat
    pyr = buf.to(DEV)
    ff, co = _pm(ffeats).to(DEV), _pm(coords).to(DEV)
    # fp64 run of the same oracle = the yardstick: at 160-pixel-wide maps a sample position carries ~1e-5 px of fp32
    # rounding from the normalise/un-normalise round trip (:318-319 + grid_sample), worth ~5e-5 on a tap -- the fp32
    # oracle itself is that far from fp64, so the kernels are held to twice the oracle's own fp32 error
    ref64 = torch.cat([O.corr_sample([p.double() for p in pyr_ref], ffeats[:, :, n0:n0 + 512].double(),
                                     coords[:, :, n0:n0 + 512].double()) for n0 in range(0, N, 512)], dim=2)
    ref_pm, ref64_pm = _pm(ref), _pm(ref64)
    Xd = ops.mixer_input_build(pyr, B, H8, W8, ff, co).cpu()
    Xt = ops.mixer_input_build_tiled(pyr, B, H8, W8, ff, co).cpu()
    floor = float((ref_pm.double() - ref64_pm).abs().max())
    e_d = float((Xd[:, 128:324].double() - ref64_pm).abs().max())
    e_t = float((Xt[:, 128:324].double() - ref64_pm).abs().max())
    print(f"config-4 geometry gather vs fp64 oracle: direct {e_d:.2e}, tiled {e_t:.2e}, fp32 oracle {floor:.2e} "
          f"(|corr| ~ {float(ref_pm.abs().max()):.1f})")
    assert floor < 1e-4 and e_d < 2 * floor and e_t < 2 * floor
    assert float((Xd[:, 128:324] - ref_pm).abs().max()) < 2e-4 and float((Xt[:, 128:324] - ref_pm).abs().max()) < 2e-4
    assert torch.equal(Xt[8:16, 128:324], torch.zeros(8, 196))       # particle 1 sits fully outside the map at every level
    assert torch.equal(Xd[:, :128], Xt[:, :128]) and torch.equal(Xd[:, 324:], Xt[:, 324:])


def test_config4_gather_bf16_matrix_cores_vs_oracle():
    """The bf16 mode at config-4 geometry (90x160 maps, the 64x64 grid + jitter): gather_mfma_kernel against the oracle's
    CorrBlock on bf16-rounded maps and features in fp64, all 4096 x 8 x 196 taps, and against the direct bf16-map kernel."""
    from pips_amd import ops, _lib
    from oracle import pips_oracle as O
    lib = _lib.load()
    B, H8, W8, N = 1, 90, 160, 4096
    assert lib.pips_gather_route(B, N, H8, W8, 32) == 2 and lib.pips_gather_route(B, N, H8, W8, 0) == 1
    assert lib.pips_gather_route(1, 256, 46, 62, 32) == 2                  # BASELINE configs[2] (21 per tile): the bf16 mode's matrix-core kernel since round 6 ...
    assert lib.pips_gather_route(1, 256, 46, 62, 0) == 0 and lib.pips_gather_route(1, 128, 46, 62, 32) == 0      # ... the fp32 mode and sparser sets: the direct kernels
    g = torch.Generator().manual_seed(22)
    fmaps = torch.randn(B, 8, 128, H8, W8, generator=g)
    ffeats = torch.randn(B, 8, N, 128, generator=g)
    coords = (_grid(N, H8 * 8, W8 * 8) / 8.0).reshape(1, 1, N, 2).repeat(B, 8, 1, 1) + torch.randn(B, 8, N, 2, generator=g) * 1.5
    coords[0, :, 0] = torch.tensor([-6.0, 3.0])
    coords[0, :, 1] = torch.tensor([W8 + 40.0, H8 + 2.0])
    coords[0, :, 2] = torch.tensor([31.999998, 16.0])
    coords[0, :, 3] = torch.tensor([16.0, 47.999996])
    pyr_ref = O.build_pyramid(fmaps)
    pyr_bf = [p.bfloat16().double() for p in pyr_ref]
    ffb = ffeats.bfloat16().double()
    ref = torch.cat([O.corr_sample(pyr_bf, ffb[:, :, n0:n0 + 256], coords[:, :, n0:n0 + 256].double())
                     for n0 in range(0, N, 256)], dim=2)
    buf = torch.zeros(lib.pips_pyramid_floats(B * 8, H8 * 8, W8 * 8, 8))
    for l, p in enumerate(pyr_ref):
        off = lib.pips_pyramid_offset(B * 8, H8 * 8, W8 * 8, 8, l)
        buf[off:off + p.numel()] = p.reshape(B * 8, 128, p.shape[-2], p.shape[-1]).permute(0, 2, 3, 1).reshape(-1)
    pyr = ops.pyramid_mirror(buf.to(DEV), B * 8, H8 * 8, W8 * 8, 8)
    ff, co = _pm(ffeats).to(DEV), _pm(coords).to(DEV)
    Xm = ops.mixer_input_build_tiled(pyr, B, H8, W8, ff, co, bf16_maps=True).cpu()
    Xd = ops.mixer_input_build(pyr, B, H8, W8, ff, co, bf16_maps=True).cpu()
    err = float((Xm[:, 128:324].double() - _pm(ref)).abs().max())
    errd = float((Xd[:, 128:324].double() - _pm(ref)).abs().max())
    dd = float((Xm[:, 128:324] - Xd[:, 128:324]).abs().max())
    print(f"config-4 geometry, bf16 mode: matrix-core gather vs fp64 oracle on bf16 operands {err:.2e}, the direct bf16-map kernel vs the "
          f"same oracle {errd:.2e}, route against route {dd:.2e}")
    # (sample positions are fp32 here, fp64 in the yardstick: ~5e-5 at 160-pixel-wide maps; the two routes share them and the
    #  operands -- round 6 -- so they differ by summation order only)
    assert err < 1.5e-4 and errd < 1.5e-4 and dd < 1e-4
    assert torch.equal(Xm[8:16, 128:324], torch.zeros(8, 196))
    assert torch.equal(Xd[:, :128], Xm[:, :128]) and torch.equal(Xd[:, 324:], Xm[:, 324:])


def test_config4_geometry_bf16_mode_end_to_end_against_autocast_oracle(weights_tamed):
    """The bf16 mode x dense grid combination of the product path, compared for the first time (round 5): one clip of 8 x 720x1280
    frames, the 64x64 grid, I=6, ``torch.autocast`` around the module -> bf16 encoder, bf16 mixer, and the correlation gather on
    the matrix cores (pips_gather_route = 2).  Against the oracle on the same GPU under ``torch.autocast("cuda", bfloat16)`` (the way
    the reference itself would be run in bf16) and against its fp32 run: the config-3 gate, 2e-2 px on tamed weights."""
    from pips_amd import Pips, _lib
    from oracle import pips_oracle as O
    B, H, W, N = 1, 720, 1280, 4096
    assert _lib.load().pips_gather_route(B, N, H // 8, W // 8, 32) == 2
    g = torch.Generator().manual_seed(6)
    rgbs = torch.randint(0, 256, (B, 8, 3, H, W), generator=g).float().to(DEV)
    xys = (_grid(N, H, W).unsqueeze(0) + torch.rand(B, N, 2, generator=g) * 4.0).to(DEV)
    sd = {k: v.to(DEV) for k, v in weights_tamed.items()}
    with torch.no_grad():
        ref32 = [p.cpu() for p in O.forward(sd, xys, rgbs, iters=6, stride=8)[0]]
        torch.cuda.empty_cache()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = O.forward(sd, xys, rgbs, iters=6, stride=8)
        refbf, visbf = [p.float().cpu() for p in out[0]], out[2].float().cpu()
    del sd, out
    torch.cuda.empty_cache()
    m = Pips(stride=8)
    m.load_state_dict(weights_tamed)
    m = m.to(DEV).eval()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        preds, _, vis, _ = m(xys, rgbs, iters=6)
    preds = [p.cpu() for p in preds]
    assert all(torch.isfinite(p).all() for p in preds)
    e_bf = max(float((p - r).abs().max()) for p, r in zip(preds, refbf))
    e_32 = max(float((p - r).abs().max()) for p, r in zip(preds, ref32))
    e_ref = max(float((a - b).abs().max()) for a, b in zip(refbf, ref32))
    e_vis = float((vis.cpu() - visbf).abs().max())
    print(f"config-4 geometry, bf16 mode end to end (B=1, N=4096, I=6): HIP vs bf16-autocast oracle {e_bf:.2e} px (vis logits {e_vis:.2e}), "
          f"HIP vs fp32 oracle {e_32:.2e} px, autocast oracle vs fp32 oracle {e_ref:.2e} px")
    assert e_bf < 2e-2 and e_32 < 2e-2 and e_vis < 0.15


def test_bf16_mode_dense_query_set_agrees_with_its_particle_shards(weights_tamed):
    """The advisor's round-5 finding: in the bf16 mode a particle's correlations depended on WHICH gather kernel its query set
    reached (dense sets: bf16 x bf16 on the matrix cores; sparse sets: fp32 features x bf16 maps), so `dist.track_sharded_particles`
    -- which turns one dense set into G sparse ones -- changed a particle's result by 7e-3 per tap.  Round 6: both routes round the
    features to bf16 (nets/pips.py:394-397 casts both matmul operands).  One clip, a 2304-point grid on 46 x 62 maps under autocast:
    the whole set (route 2 asserted) against its sixteen shards of 144 (route 0 asserted), same cached maps.  What is left is the
    order of the fp32 sums in the gather (<= 1e-4 per tap, tests/test_kernels_gpu.py) and, from the second iteration on, what the
    bf16 mixer makes of a tap that rounds the other way (its in-projection rounds X to bf16: already in the first iterate): printed, gated at
    half of the bf16 mode's own 2e-2 px budget."""
    from pips_amd import Pips, _lib
    lib = _lib.load()
    B, H, W, N, G = 1, 368, 496, 2304, 16                     # a 48 x 48 grid, 144 queries per shard
    assert lib.pips_gather_route(B, N, H // 8, W // 8, 32) == 2 and lib.pips_gather_route(B, N // G, H // 8, W // 8, 32) == 0
    g = torch.Generator().manual_seed(9)
    rgbs = torch.randint(0, 256, (B, 8, 3, H, W), generator=g).float().to(DEV)
    xys = (_grid(N, H, W).unsqueeze(0) + torch.rand(B, N, 2, generator=g) * 2.0).to(DEV)
    m = Pips(stride=8)
    m.load_state_dict(weights_tamed)
    m = m.to(DEV).eval()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        cache = m.encode(rgbs)
        assert cache.bf16_maps
        whole = m.track(cache, xys, iters=6)
        parts = [m.track(cache, xys[:, r * (N // G):(r + 1) * (N // G)].contiguous(), iters=6) for r in range(G)]
    e1 = float((whole[0][0] - torch.cat([p[0][0] for p in parts], dim=2)).abs().max())
    e6 = float((whole[0][-1] - torch.cat([p[0][-1] for p in parts], dim=2)).abs().max())
    ev = float((whole[2] - torch.cat([p[2] for p in parts], dim=2)).abs().max())
    print(f"bf16 mode, dense set (matrix-core gather) vs its 16 particle shards (direct gather): first iterate {e1:.2e} px, after 6 "
          f"iterations {e6:.2e} px, vis logits {ev:.2e}")
    assert e1 < 4e-3 and e6 < 1e-2 and ev < 5e-2          # measured 1.2e-3 / 4.5e-3 / 7.5e-3 (profiles/r6_bf16_parity_tests.log)


def test_config4_teacher_forced_iteration(weights_raw):
    """B=1, 8 x 720x1280 frames, N=4096 grid, one update iteration through the product path (encoder -> tiled gather
    -> mixer -> update).  Particles are independent given the maps, so the oracle (CPU) runs a 256-particle subset."""
    from pips_amd import Pips
    from oracle import pips_oracle as O
    H, W, N = 720, 1280, 4096
    g = torch.Generator().manual_seed(2)
    rgbs = torch.randint(0, 256, (1, 8, 3, H, W), generator=g).float()
    xys = _grid(N, H, W).unsqueeze(0)
    sub = torch.randperm(N, generator=g)[:256]
    ref_p, _, ref_vis, ref_ff = O.forward(weights_raw, xys[:, sub], rgbs, iters=1, stride=8)
    m = Pips(stride=8)
    m.load_state_dict(weights_raw)
    m = m.to(DEV).eval()
    preds, _, vis, ffeat, _ = m(xys.to(DEV), rgbs.to(DEV), iters=1, return_feat=True)
    err = float((preds[0].cpu()[:, :, sub] - ref_p[0]).abs().max())
    verr = float((vis.cpu()[:, :, sub] - ref_vis).abs().max())
    print(f"config-4 geometry, first iterate (raw weights): max |dtraj| = {err:.2e} px, |dvis| = {verr:.2e}")
    assert err < 1e-3 and verr < 1e-3
    assert float((ffeat.cpu()[:, sub] - ref_ff).abs().max()) < 2e-4


def test_config4_end_to_end_b4_i6_against_oracle_on_device(weights_tamed):
    """BASELINE configs[3] whole, the geometry bench.py's config4 leg times: B=4 clips of 8 x 720x1280 frames, the 64x64 query
    grid (N=4096), I=6 -- every one of the 131 072 tracks through the product path (encoder -> tiled gather -> mixer -> update)
    against the oracle run on the SAME GPU with torch-ROCm fp32 ops (MIOpen convolutions, rocBLAS matmuls, the dense correlation
    volumes materialised: 7.5 GB at level 0).  The oracle is the checker here, not the product; tamed weights, whose own fp32
    noise floor over six iterations is 8e-5 px (tests/test_forward_gpu.py docstring), gate 1e-3 px."""
    from pips_amd import Pips
    from oracle import pips_oracle as O
    B, H, W, N = 4, 720, 1280, 4096
    g = torch.Generator().manual_seed(5)
    rgbs = torch.randint(0, 256, (B, 8, 3, H, W), generator=g).float().to(DEV)
    xys = (_grid(N, H, W).unsqueeze(0).repeat(B, 1, 1) + torch.rand(B, N, 2, generator=g) * 4.0).to(DEV)
    sd = {k: v.to(DEV) for k, v in weights_tamed.items()}
    with torch.no_grad():
        ref_p, _, ref_vis, ref_ff = O.forward(sd, xys, rgbs, iters=6, stride=8)
    ref_p = [p.cpu() for p in ref_p]
    ref_vis, ref_ff = ref_vis.cpu(), ref_ff.cpu()
    del sd
    torch.cuda.empty_cache()
    m = Pips(stride=8)
    m.load_state_dict(weights_tamed)
    m = m.to(DEV).eval()
    moved = float((ref_p[-1] - xys.cpu().unsqueeze(1)).abs().max())
    # exact fp32 MFMA, then the fp32-grade split-bf16 matrix mode (config 4 is 77 % fp32 GEMM: the mode that may exceed the fp32
    # roof must meet the same gate at the same size) -- one oracle run serves both
    for matmul in ("exact", "split"):
        m.matmul = matmul
        preds, _, vis, ffeat, _ = m(xys, rgbs, iters=6, return_feat=True)
        err = [float((p.cpu() - r).abs().max()) for p, r in zip(preds, ref_p)]
        verr = float((vis.cpu() - ref_vis).abs().max())
        print(f"config 4 end to end (B=4, N=4096, I=6), matmul={matmul}: per-iteration max |dtraj| px {['%.1e' % e for e in err]}, "
              f"|dvis| {verr:.1e}, tracks move up to {moved:.2f} px")
        assert max(err) < 1e-3 and verr < 1e-3
        assert float((ffeat.cpu() - ref_ff).abs().max()) < 2e-4
        del preds, vis, ffeat
        torch.cuda.empty_cache()


def test_config5_chained_tracking_stride4(weights_tamed):
    """chain_demo.py geometry: 360x640 frames, stride 4, S=8 windows chained on visibility over T=24 frames, N=64
    particles, against the loop restatement (frame maps cached in the oracle too: oracle/chain_oracle.py)."""
    from pips_amd import Pips, drivers
    from oracle import chain_oracle
    g = torch.Generator().manual_seed(5)
    T, H, W, N = 24, 360, 640, 64
    base = torch.randint(0, 256, (1, 1, 3, H, W), generator=g).float()
    video = torch.cat([(base * (1 - 0.02 * t) + 5.0 * t).clamp(0, 255).round() for t in range(T)], dim=1)
    video = (video + torch.randint(0, 30, video.shape, generator=g).float()).clamp(0, 255)
    xy0 = _grid(N, H, W, margin=16.0).unsqueeze(0)
    ref, hops = chain_oracle.chain(weights_tamed, video, xy0, iters=6, stride=4, cache_frames=True)
    m = Pips(stride=4)
    m.load_state_dict(weights_tamed)
    m = m.to(DEV).eval()
    got = drivers.track_chained(m, video.to(DEV), xy0.to(DEV), iters=6).cpu()
    err = float((got - ref).abs().max())
    print("config-5 geometry chained: max |dtraj| px", err, " hops/particle", sum(len(h) for h in hops) / N)
    assert tuple(got.shape) == (1, T, N, 2) and err < 1e-3


def test_config5_full_size_t100_n256_against_oracle_on_device(weights_tamed):
    """BASELINE configs[4] at its OWN size (chain_demo.py:40-83): 100 frames of 360x640, stride 4, N = 256 (a 16x16 grid at
    frame 0), I = 6, S = 8 windows chained on visibility -- ``drivers.track_chained`` against the chaining oracle run on the
    SAME GPU with torch-ROCm fp32 ops (oracle/chain_oracle.chain_lockstep: the loop of ``chain``, which is pinned to the
    reference's own text, with every particle as one clip of the oracle forward; held to ``chain`` on the CPU by
    tests/test_oracle_golden.py::test_chain_lockstep_equals_chain).  Gate: identical hop sequences for every particle and
    1e-3 px on all 100 x 256 positions (tamed weights)."""
    from pips_amd import Pips, drivers
    from oracle import chain_oracle
    g = torch.Generator().manual_seed(9)
    T, H, W, N = 100, 360, 640, 256
    base = torch.randint(0, 256, (1, 1, 3, H, W), generator=g).float()
    video = torch.cat([(base * (1 - 0.005 * t) + 1.2 * t).clamp(0, 255).round() for t in range(T)], dim=1)
    video = (video + torch.randint(0, 30, video.shape, generator=g).float()).clamp(0, 255).to(DEV)
    xy0 = _grid(N, H, W, margin=16.0).unsqueeze(0).to(DEV)
    sd = {k: v.to(DEV) for k, v in weights_tamed.items()}
    ref, ref_hops = chain_oracle.chain_lockstep(sd, video, xy0, iters=6, stride=4)
    ref = ref.cpu()
    del sd
    torch.cuda.empty_cache()
    m = Pips(stride=4)
    m.load_state_dict(weights_tamed)
    m = m.to(DEV).eval()
    got, hops = drivers.track_chained(m, video, xy0, iters=6, return_hops=True)
    got = got.cpu()
    err = float((got - ref).abs().max())
    nh = sum(len(h) for h in ref_hops)
    diff = [n for n in range(N) if hops[n] != ref_hops[n]]
    print(f"config 5 at full size (T=100, N=256, 360x640 s4, I=6): max |dtraj| {err:.2e} px over {T * N} positions, "
          f"{nh} window forwards ({nh / N:.1f} hops per particle), particles with a different hop sequence: {len(diff)}")
    assert tuple(got.shape) == (1, T, N, 2)
    assert not diff
    assert err < 1e-3
