#!/usr/bin/env python
"""bench.py -- particle-updates/s of the PIPs inference hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config 2|3] [--no-extras] [--no-cpu-baseline]

One "step" = one whole ``Pips.forward`` (encoder + 6 update iterations) over one batch of
synthetic clips already resident in HBM.  Workload = BASELINE.json configs[1]:
B=1 clip per GPU, S=8 frames, 368x496, N=256 particles, I=6 iterations, fp32, stride 8,
seeded random-init weights (the reference checkpoint is not obtainable offline).

--gpus N > 1: one process per GPU, clips sharded on the batch axis (weak scaling), one RCCL
all-gather of the final [x,y,vis] per step.  Launched by ``python -m torch.distributed.run``
the script reads RANK/LOCAL_RANK/WORLD_SIZE; launched bare (WORLD_SIZE unset) it re-executes
itself under torch.distributed.run with N ranks.  WORLD_SIZE != N is an error.

Prints ONE JSON line (rank 0): the driver contract plus
  "ms_per_step_median"  -- median of the per-step durations between HIP events recorded around every timed step (the
                           contract's ms_per_step / value stay the wall-clock bracket over all K steps);
  "roofline"            -- dominant kernel (fp32-MFMA GEMM of the mixer) vs the 157.3 TF fp32 peak,
                           timed live with HIP events on the launch stream (any number of ranks: measured on rank 0);
  "config3"             -- BASELINE configs[2] (bf16 operands): "weak" = B=8 clips per GPU, "strong" = B=64 clips in total
                           split over the ranks, both barrier-bracketed, max over ranks, all-gather included;
  "collective_ms"       -- the final all-gather of [x,y,vis] alone (N > 1), HIP-event timed;
  "gather"              -- the correlation-gather kernel at config 2 (L2-resident; informational);
  "stages_ms"           -- where one forward's time goes;
  "config3"/"config4"/"config5" -- the other BASELINE configs, each with its own numbers; config4
                           carries the HBM roofline of the correlation gather (SURVEY.md 8(d)(i) bytes);
  "torch_rocm_baseline" -- the same forward through stock PyTorch-ROCm ops (oracle restatement on device
                           tensors: MIOpen/rocBLAS) on this GPU -- the same-hardware comparator;
  "cpu_baseline"        -- the CPU oracle (a port of the reference forward) on this host, N=1 only.
"""
from __future__ import annotations

import argparse
import json
import os
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

S, H, W, NPTS, ITERS, STRIDE = 8, 368, 496, 256, 6, 8
PEAK_F32_MFMA_TF = 157.3           # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_BF16_MFMA_TF = 2500.0         # MI355X_MICROARCH.md: ~2.5 PF dense bf16 MFMA (16x the fp32 MFMA rate)
PEAK_HBM_GBS = 8000.0              # spec; 6290 measured-achievable
MIXER_STREAM = None                # --mixer-stream f32|bf16: the bf16 legs' residual-stream type (None: the module's default)


def respawn_under_launcher(gpus):
    """Bare ``python bench.py --gpus N`` (N > 1): run N ranks of this script on this node."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def make_inputs(rank, device, b, h=H, w=W, n=NPTS):
    import torch
    g = torch.Generator().manual_seed(1 + rank)
    rgbs = torch.randint(0, 256, (b, S, 3, h, w), generator=g).float()
    xys = torch.rand(b, n, 2, generator=g) * torch.tensor([w - 1.0, h - 1.0])
    return xys.to(device), rgbs.to(device)


def ev_time_ms(fn, reps):
    """Average milliseconds of fn() over reps calls, HIP events on the current stream."""
    import torch
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps


def gemm_roofline(model, device, b, flags=0):
    """The dominant kernel -- the mixer's channel-mix GEMMs -- timed IN SITU: a real mixer pass on the real weights with a
    HIP event pair around every GEMM launch on the launch stream (pips_mixer_fwd_timed); mean over the 12 layers and 5
    passes, empty-pair overhead subtracted.  Returns {name: {ms, tflops}}."""
    import torch
    from pips_amd import ops
    arena = model._packed(device, need=7)
    M = b * NPTS * S
    X = torch.randn(M, 544, generator=torch.Generator().manual_seed(0)).to(device)
    ups, downs, ovh = [], [], []
    for _ in range(6):
        _, t = ops.mixer_fwd_timed(arena, X, flags=flags)
        ups.append(t["up_proj"]); downs.append(t["down_proj"]); ovh.append(t["event_overhead"])
    ups, downs, ovh = ups[1:], downs[1:], ovh[1:]
    o = sum(ovh) / len(ovh)
    t_up, t_down = sum(ups) / len(ups) - o, sum(downs) / len(downs) - o
    flops = 2.0 * M * 2048 * 512
    tr = gemm_trains(arena, X, flags)
    return {"up_proj(M=%d,N=2048,K=512)" % M: {"ms": tr["up_proj"], "tflops": flops / tr["up_proj"] / 1e9, "kernel_body_ms": t_up},
            "down_proj(M=%d,N=512,K=2048)" % M: {"ms": tr["down_proj"], "tflops": flops / tr["down_proj"] / 1e9,
                                                  "kernel_body_ms": t_down}}


def gemm_trains(arena, X, flags=0):
    """{up_proj, down_proj} ms per launch from launch TRAINS (ops.mixer_gemm_train: the pass's 12 up- / 12 down-projections back
    to back between one HIP event pair, start to start); median of 5 trains of 48 launches."""
    from pips_amd import ops
    runs = [ops.mixer_gemm_train(arena, X, flags=flags, reps=4) for _ in range(6)][1:]
    return {k: statistics.median(r[k] for r in runs) for k in ("up_proj", "down_proj")}


def gemm_kernel_name(M, N, K, epi):
    """The kernel a channel-mix Linear of this shape runs on (pips_gemm_f32_route: gemm_f32_t4.hip's assembly kernels or igemm_f32_kernel)."""
    from pips_amd import _lib
    route = _lib.load().pips_gemm_f32_route(M, N, K, epi)
    if route == 1:
        return "gemm_f32_t4u_kernel<%d>" % (0 if epi == 1 else 1)
    if route == 2:
        return "gemm_f32_t4e_kernel"
    return "igemm_f32_kernel<64, 64, 2, 2, %d, false>" % (1 if epi == 1 else 2)


def rocprof_avg_us(kernel_substr, stats_glob="r*_kernel_stats.txt"):
    """Average duration (us) of a kernel in the newest committed rocprofv3 kernel-trace summary of the headline command
    (profiles/rN_kernel_stats.txt, tools/profile_round.sh).  Returns (us or None, source)."""
    import glob
    import re
    files = [f for f in glob.glob(os.path.join(ROOT, "profiles", stats_glob)) if re.search(r"r\d+_kernel_stats\.txt$", f)]
    files.sort(key=lambda f: int(re.search(r"r(\d+)_kernel_stats", f).group(1)))
    if not files:
        return None, None
    try:
        for line in open(files[-1]):
            if kernel_substr in line:
                cols = line.rsplit(None, 4)                       # kernel | calls | total_us | avg_us | pct
                return float(cols[-2]), "profiles/" + os.path.basename(files[-1])
    except (OSError, ValueError, IndexError):
        pass
    return None, None


def roofline_object(kern, fake=False):
    """The dominant kernel = the slower of the two channel-mix GEMM shapes.  `frac` comes from LAUNCH TRAINS (start-to-start
    duration of back-to-back launches: what the forward pays per launch and what a rocprofv3 kernel trace of the forward shows);
    the event-pair-minus-overhead figure of rounds 1-4 (the kernel body without its launch boundary) stays beside it as
    kernel_body_ms, and the committed rocprofv3 summary's average as launch_ms_rocprof / frac_rocprof."""
    dom = max(kern.items(), key=lambda kv: kv[1]["ms"])
    # HBM-side bytes per launch of that kernel: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this very
    # command, committed per round under profiles/ (a live bench run cannot collect PMC counters itself)
    M = int(dom[0].split("M=")[1].split(",")[0])
    name = gemm_kernel_name(M, 2048, 512, 1) if dom[0].startswith("up") else gemm_kernel_name(M, 512, 2048, 2)
    traffic, traffic_src = pmc_traffic(name)
    flops = 2.0 * M * 2048 * 512
    rp_us, rp_src = rocprof_avg_us(name.split("(")[0])
    out = {"bound": "mfma", "achieved": dom[1]["tflops"], "peak": PEAK_F32_MFMA_TF,
           "unit": "TFLOP/s", "frac": dom[1]["tflops"] / PEAK_F32_MFMA_TF, "traffic": traffic,
           "traffic_source": traffic_src,
           "traffic_note": "2 x FETCH_SIZE + WRITE_SIZE at the fabric: FETCH_SIZE also counts Infinity-Cache hits (the 104 MB of "
                           "weights live there), so bytes above the algorithmic count are mostly cache hits, not HBM re-reads",
           "algorithmic_flop_per_launch": flops,
           "kernel": name + " " + dom[0], "launch_ms": dom[1]["ms"],
           "timing": "launch train: the 12 layers' launches of this shape back to back between ONE HIP event pair on the launch "
                     "stream, x4, median of 5 trains; start-to-start, nothing subtracted (pips_mixer_gemm_train)",
           "kernel_body_ms": dom[1].get("kernel_body_ms"),
           "kernel_body_timing": "HIP event pair around each in-situ launch of a mixer pass, empty-pair overhead subtracted "
                                 "(rounds 1-4 reported this as launch_ms; it leaves the launch boundary out)",
           "launch_ms_rocprof": None if rp_us is None else rp_us / 1e3,
           "frac_rocprof": None if rp_us is None else flops / (rp_us * 1e-6) / 1e12 / PEAK_F32_MFMA_TF,
           "rocprof_source": rp_src,
           "all": kern}
    return out


def stage_profile(model, xys, rgbs, device, b):
    """Per-stage and per-kernel timings with the staged C-ABI entry points (same kernels)."""
    import torch
    from pips_amd import ops
    arena = model._packed(device, need=7)          # this leg also times the bf16 / split kernels on the same arena
    F = b * S
    H8, W8 = H // STRIDE, W // STRIDE
    M = b * NPTS * S
    frames = rgbs.reshape(F, 3, H, W)
    pyr = ops.encoder_fwd(arena, frames, STRIDE)
    out = {}
    out["encoder"] = min(ev_time_ms(lambda: ops.encoder_fwd(arena, frames, STRIDE), 3) for _ in range(3))
    g = torch.Generator().manual_seed(0)
    ffeats = torch.randn(M, 128, generator=g).to(device)
    coords = (torch.rand(M, 2, generator=g) * torch.tensor([W8 - 1.0, H8 - 1.0])).to(device)
    X = ops.mixer_input_build(pyr, b, H8, W8, ffeats, coords)
    t_gather = ev_time_ms(lambda: ops.mixer_input_build(pyr, b, H8, W8, ffeats, coords), 20)
    out["mixer_input(gather)"] = t_gather * ITERS
    delta = ops.mixer_fwd(arena, X)
    out["mixer"] = ev_time_ms(lambda: ops.mixer_fwd(arena, X), 5) * ITERS
    c0 = coords.clone()
    out["state_update"] = ev_time_ms(
        lambda: ops.state_update(arena, delta, ffeats, coords, c0, b, NPTS, float(STRIDE)), 10) * ITERS

    # dominant kernel: the channel-mix GEMMs (igemm_f32_kernel), timed IN SITU: a real mixer pass
    # on the real weights/activations with a HIP event pair around every GEMM launch on the
    # launch stream (pips_mixer_fwd_timed); mean over the 12 layers and 5 passes.
    def timed(flags):
        ups, downs, ovh = [], [], []
        for _ in range(5):
            _, t = ops.mixer_fwd_timed(arena, X, flags=flags)
            ups.append(t["up_proj"]); downs.append(t["down_proj"]); ovh.append(t["event_overhead"])
        o = sum(ovh) / len(ovh)      # an event pair costs a marker-to-marker gap even with nothing between
        return sum(ups) / len(ups) - o, sum(downs) / len(downs) - o
    t_up, t_down = timed(0)
    flops = 2.0 * M * 2048 * 512
    tr = gemm_trains(arena, X, 0)                       # start-to-start launch durations: what roofline.frac is quoted on
    kern = {
        "up_proj(M=%d,N=2048,K=512)" % M: {"ms": tr["up_proj"], "tflops": flops / tr["up_proj"] / 1e9, "kernel_body_ms": t_up},
        "down_proj(M=%d,N=512,K=2048)" % M: {"ms": tr["down_proj"], "tflops": flops / tr["down_proj"] / 1e9,
                                              "kernel_body_ms": t_down},
    }
    s_up, s_down = timed(16)
    split = {
        "up_proj_ms": s_up, "down_proj_ms": s_down,
        "fp32_equiv_tflops": {"up_proj": flops / s_up / 1e9, "down_proj": flops / s_down / 1e9},
        "bf16_mfma_tflops_issued": 6 * flops / max(s_up, s_down) / 1e9,
        "frac_of_bf16_peak": 6 * flops / max(s_up, s_down) / 1e9 / PEAK_BF16_MFMA_TF,
        "mixer_ms": ev_time_ms(lambda: ops.mixer_fwd(arena, X, split=True), 5) * ITERS,
        "encoder_ms": min(ev_time_ms(lambda: ops.encoder_fwd(arena, frames, STRIDE, split=True), 3) for _ in range(3)),
    }
    lv = sum((H8 >> l) * (W8 >> l) for l in range(4))
    comp_bytes = b * S * (lv * 128 * 4 + NPTS * 128 * 4 + NPTS * 8 + NPTS * 196 * 4)
    gather = {
        "kernel": "mixer_input_kernel", "ms_per_launch": t_gather,
        "compulsory_GBs": comp_bytes / t_gather / 1e6,
        "note": "config 2's 18 MB footprint is L2/MALL-resident: launch-latency-bound; the HBM roofline of the "
                "gather is reported at config 4 (config4.gather_roofline)",
    }
    return out, kern, gather, split


def pmc_traffic(kernel_substr):
    """HBM-side bytes per launch of a kernel from this round's committed rocprofv3 --pmc passes (FETCH_SIZE and
    WRITE_SIZE in separate runs of the same commands, tools/profile_round.sh; a live bench run cannot collect PMC
    counters itself).  Returns (bytes or None, source)."""
    import glob
    import re
    files = [f for f in glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")) if re.search(r"r\d+_pmc_traffic\.json$", f)]
    files.sort(key=lambda f: int(re.search(r"r(\d+)_pmc_traffic", f).group(1)))           # the latest round's passes (r10 after r9)
    name = os.path.basename(files[-1]) if files else "r2_pmc_traffic.json"
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", name)))["kernels"]
        for k, v in pmc.items():
            if kernel_substr in k:
                return v["hbm_bytes"], "profiles/" + name
    except (OSError, KeyError, ValueError):
        pass
    return None, None


def _max_over_ranks(dt, world, device, on_device):
    if world == 1:
        return dt
    import torch
    import torch.distributed as dist
    t = torch.tensor([dt], device=device if on_device else "cpu", dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def config3_leg(device, world=1, rank=0, fake=False, on_device_collectives=True):
    """BASELINE configs[2]: bf16 MFMA operands, clips sharded over the ranks + one all-gather of [x,y,vis] per step.
    weak = 8 clips per GPU (the config's per-GPU share at 8 GPUs); strong = 64 clips in total, 64 / world per GPU, run as
    forwards of <= 8 clips.  Every figure: barrier + synchronize on both sides, max over ranks, >= 10 timed steps."""
    import torch
    import torch.distributed as dist
    from pips_amd import dist as pdist
    b = 8
    if fake:
        model = _CpuStandIn()
        xys, rgbs = make_inputs(rank, device, b, 32, 32, 16)
        npts = 16
    else:
        from pips_amd import Pips
        model = Pips(S=S, stride=STRIDE).to(device).eval()
        model.mixer_dtype = model.encoder_dtype = torch.bfloat16
        if MIXER_STREAM is not None:                            # (tuning: --mixer-stream; the default is the module's own)
            model.mixer_stream_dtype = torch.bfloat16 if MIXER_STREAM == "bf16" else torch.float32
        xys, rgbs = make_inputs(rank, device, b)
        npts = NPTS
    sync = (lambda: None) if fake else torch.cuda.synchronize

    def step(nclips):
        # forwards of <= 8 clips (one GPU's share of the weak case) over DISTINCT clips; every chunk's result is kept and
        # the step ends with ONE all-gather of all nclips clips of this rank (the real payload of the exchange)
        outs = []
        for c0 in range(0, nclips, b):
            n = min(b, nclips - c0)
            preds, _, vis, _ = model(xys_all[c0:c0 + n], rgbs_all[c0:c0 + n], iters=ITERS)
            outs.append((preds[-1], vis))
        if world > 1:
            tr = outs[0][0] if len(outs) == 1 else torch.cat([o[0] for o in outs], dim=0)
            vi = outs[0][1] if len(outs) == 1 else torch.cat([o[1] for o in outs], dim=0)
            pdist.all_gather_result(tr, vi)

    def timed(nclips, reps):
        for _ in range(2):
            step(nclips)
        if world > 1:
            dist.barrier()
        sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            step(nclips)
        sync()
        if world > 1:
            dist.barrier()
        dt = _max_over_ranks(time.perf_counter() - t0, world, device, on_device_collectives and not fake)
        return dt / reps * 1e3

    flop_per_update = 72.2e6
    xys_all, rgbs_all = xys, rgbs
    t_weak = timed(b, 10)
    ups_weak = world * b * S * npts * ITERS / t_weak * 1e3
    total_strong = 64
    out = {"workload": "BASELINE configs[2]: S=8 368x496 N=256 I=6, bf16 MFMA operands and bf16 encoder activations (fp32 "
                       "accumulate / residual stream / state), encoder included, clips sharded over the ranks",
           "n_gpus": world, "unit": "particle-updates/s", "dtype": "bf16 MFMA operands",
           "weak": {"clips_per_gpu": b, "clips_total": b * world, "ms_per_step": t_weak, "value": ups_weak, "steps": 10},
           # back-compatible top-level fields = the weak case (one GPU's share B=8)
           "value": ups_weak, "ms_per_step": t_weak}
    if total_strong % world == 0:
        per = total_strong // world
        reps = 10 if per <= b else 3
        if per > b:                                             # distinct clips for every chunk of the step (1.1 GB at world = 1)
            xys_all = rgbs_all = xys = rgbs = None
            xys_all, rgbs_all = make_inputs(rank, device, per, 32, 32, 16) if fake else make_inputs(rank, device, per)
        t_strong = timed(per, reps)
        out["strong"] = {"clips_total": total_strong, "clips_per_gpu": per, "ms_per_step": t_strong,
                         "value": total_strong * S * npts * ITERS / t_strong * 1e3, "steps": reps,
                         "note": "64 distinct clips per step in forwards of <= 8 clips per GPU; one all-gather of all of a rank's clips per step"}
    per_gpu = ups_weak / world
    out["roofline"] = {"bound": "mfma", "achieved": per_gpu * flop_per_update / 1e12, "peak": PEAK_BF16_MFMA_TF, "unit": "TFLOP/s",
                       "frac": per_gpu * flop_per_update / 1e12 / PEAK_BF16_MFMA_TF,
                       "note": "whole forward per GPU: 72.2 MFLOP per particle-update (SURVEY 8d) against the dense bf16 MFMA peak"}
    return out


def collective_leg(device, world, fake, npts):
    """The data path's only exchange, alone: all_gather_into_tensor of the packed [x,y,vis] of 8 clips per rank
    (196 KB per rank at N=256), event-timed (wall-clock under the CPU stand-in), median of 20."""
    import torch
    from pips_amd import dist as pdist
    tr = torch.zeros(8, S, npts, 2, device=device)
    vi = torch.zeros(8, S, npts, device=device)
    ts = []
    for i in range(25):
        if fake:
            t0 = time.perf_counter()
            pdist.all_gather_result(tr, vi)
            dt = (time.perf_counter() - t0) * 1e3
        else:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            pdist.all_gather_result(tr, vi)
            e1.record()
            e1.synchronize()
            dt = e0.elapsed_time(e1)
        if i >= 5:
            ts.append(dt)
    return {"median_ms": statistics.median(ts), "bytes_per_rank": int(tr.numel() * 4 + vi.numel() * 4), "reps": len(ts),
            "what": "pack [x,y,vis] + all_gather_into_tensor + unpack (pips_amd.dist.all_gather_result)"}


def config4_leg(device):
    """BASELINE configs[3]: B=4 S=8 720x1280, N=4096 on a 64x64 grid, I=6, fp32.  The HBM-bound kernel of
    the config is the correlation gather: timed per launch with HIP events around the kernel itself,
    against SURVEY.md 8(d)(i) compulsory bytes (pyramid + ffeats + coords + fcorrs = 483.9 MB)."""
    import torch
    from pips_amd import Pips, ops
    b, h, w, n = 4, 720, 1280, 4096
    model = Pips(S=S, stride=STRIDE).to(device).eval()
    g = torch.Generator().manual_seed(1)
    rgbs = torch.randint(0, 256, (b, S, 3, h, w), generator=g, dtype=torch.uint8).to(device).float()
    k = int(round(n ** 0.5))
    gy, gx = torch.meshgrid(torch.linspace(8, h - 8, k), torch.linspace(8, w - 8, k), indexing="ij")
    xys = torch.stack([gx.reshape(-1), gy.reshape(-1)], -1).unsqueeze(0).repeat(b, 1, 1).to(device)
    fwd = lambda: model(xys, rgbs, iters=ITERS)
    preds = fwd()[0]
    t_fwd = ev_time_ms(fwd, 2)
    # the same forward in the fp32-grade split-bf16 matrix mode (config 4 is 77 % fp32 GEMM at 0.85 of a roof that mode may exceed)
    model.matmul = "split"
    fwd()
    t_split = ev_time_ms(fwd, 2)
    model.matmul = "exact"
    # gather in isolation on the real maps: iteration-0 state (the dense grid) and the state after 6 updates
    arena = model._packed(device)
    H8, W8, F, M = h // STRIDE, w // STRIDE, b * S, b * n * S
    pyr = ops.encoder_fwd(arena, rgbs.reshape(F, 3, h, w), STRIDE)
    ffeats = torch.randn(M, 128, generator=g).to(device)        # (values do not influence the timing)
    lv = sum((H8 >> l) * (W8 >> l) for l in range(4))
    comp = F * (lv * 512 + n * 512 + n * 8 + n * 196 * 4)                          # 483.9 MB
    out = {}
    for name, coords_bsn in (("iter0_grid", (xys / STRIDE).unsqueeze(1).expand(b, S, n, 2)), ("after_6_updates", preds[-1] / STRIDE)):
        c = coords_bsn.permute(0, 2, 1, 3).reshape(M, 2).contiguous()
        ts = {"bin": [], "embed": [], "gather": []}
        for i in range(12):
            _, t = ops.mixer_input_build_tiled_timed(pyr, b, H8, W8, ffeats, c)
            if i >= 2:
                for kk in ts:
                    ts[kk].append(t[kk])
        tg = statistics.mean(ts["gather"])
        t_direct = ev_time_ms(lambda: ops.mixer_input_build(pyr, b, H8, W8, ffeats, c), 3)
        out[name] = {"gather_tiled_kernel_ms": tg, "bin_particles_kernel_ms": statistics.mean(ts["bin"]),
                     "embed_rows_kernel_ms": statistics.mean(ts["embed"]), "direct_kernel_ms(mixer_input_kernel)": t_direct,
                     "achieved_GBs": comp / tg / 1e6, "frac_of_8TBs": comp / tg / 1e6 / PEAK_HBM_GBS}
    dom = out["iter0_grid"]
    # the bf16 mode of the same query set (PIPS_FLAG_BF16_MAPS: bf16 features x bf16 maps, the reference's arithmetic under autocast):
    # gather_mfma_kernel on the bf16 mirror of the same maps, against ITS algorithmic bytes (bf16 pyramid + bf16
    # features + coordinates + fp32 fcorrs = 293.8 MB)
    ops.pyramid_mirror(pyr, F, h, w, STRIDE)
    comp_bf = F * (lv * 256 + n * 256 + n * 8 + n * 196 * 4)
    c0 = (xys / STRIDE).unsqueeze(1).expand(b, S, n, 2).permute(0, 2, 1, 3).reshape(M, 2).contiguous()
    tb = {"bin": [], "embed": [], "gather": []}
    for i in range(12):
        _, t = ops.mixer_input_build_tiled_timed(pyr, b, H8, W8, ffeats, c0, bf16_maps=True)
        if i >= 2:
            for kk in tb:
                tb[kk].append(t[kk])
    tgb = statistics.mean(tb["gather"])
    path_b = sum(statistics.mean(tb[kk]) for kk in tb)
    traffic_b, traffic_b_src = pmc_traffic("gather_mfma_kernel")
    # what bounds THIS formulation on the chip (profiles/r6_probe_gather_mfma2.txt): the work items fetch 780 MB of pixel blocks through the
    # compute units' vector L1s -- 17-18 B/clk/CU = 70 us whatever the staging (registers or LDS-DMA, two or three buffers) -- and the window
    # scatter occupies the LDS store path for 192 ds_write_b32 x 4 clk per step
    fetch_floor_ms = 0.070
    gather_bf16 = {"bound": "hbm", "kernel": "gather_mfma_kernel (bf16 mode: PIPS_FLAG_BF16_MAPS on a dense query set)", "launch_ms": tgb,
                   "achieved": comp_bf / tgb / 1e6, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": comp_bf / tgb / 1e6 / PEAK_HBM_GBS,
                   "traffic": traffic_b, "traffic_source": traffic_b_src, "algorithmic_bytes_per_launch": comp_bf,
                   # the whole tiled path of an iteration: binning + embedding rows + the gather itself, and the fraction on IT
                   "gather_path_ms": path_b, "bin_particles_kernel_ms": statistics.mean(tb["bin"]),
                   "embed_rows_kernel_ms": statistics.mean(tb["embed"]), "frac_of_path": comp_bf / path_b / 1e6 / PEAK_HBM_GBS,
                   # the formulation's own floor as a fraction of 8 TB/s: what 0.80 would have to get past
                   "l1_fetch_floor_ms": fetch_floor_ms, "ceiling_frac": comp_bf / fetch_floor_ms / 1e6 / PEAK_HBM_GBS,
                   "ceiling_note": "the tile regions of the four levels are 780 MB per launch through the vector L1s at the 17-18 B/clk a "
                                   "compute unit fetches from L2 / Infinity Cache (measured: the request stream alone, registers or LDS-DMA, "
                                   "profiles/r6_probe_gather_mfma2.txt)",
                   "timing": "HIP event pair around the kernel launch, mean of 10 launches on the real maps (iteration-0 grid)"}
    # the bf16 mode END TO END at this config: the whole forward under autocast (bf16 encoder + mixer, gather_mfma_kernel)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        fwd()
        t_bf16 = ev_time_ms(fwd, 2)
    traffic, traffic_src = pmc_traffic("gather_tiled_kernel")
    valu_floor_ms = b * S * n * 4 * 64 * 128 / (105.0 * 256 * 2.4e9) * 1e3
    lds_floor_ms = 0.78 * b * S * n * 4 * 64 * 128 * 4 / (256.0 * 256 * 2.4e9) * 1e3
    path_f = dom["gather_tiled_kernel_ms"] + dom["bin_particles_kernel_ms"] + dom["embed_rows_kernel_ms"]
    return {"workload": "BASELINE configs[3]: B=4 S=8 720x1280 N=4096 (64x64 grid) I=6 fp32 stride 8, encoder included",
            "value": b * S * n * ITERS / t_fwd * 1e3, "unit": "particle-updates/s", "ms_per_step": t_fwd, "dtype": "f32",
            "split_bf16": {"ms_per_step": t_split, "value": b * S * n * ITERS / t_split * 1e3,
                           "note": "Pips.matmul='split' at the same size; same 1e-3 px gate (tests/test_config45_gpu.py)"},
            "bf16": {"ms_per_step": t_bf16, "value": b * S * n * ITERS / t_bf16 * 1e3, "dtype": "bf16 MFMA operands",
                     "note": "the same forward under torch.autocast(bfloat16): bf16 encoder and mixer, the gather on the matrix cores; "
                             "2e-2 px gate against the autocast oracle at this geometry (tests/test_config45_gpu.py)"},
            "gather_roofline": {"bound": "hbm", "kernel": "gather_tiled_kernel", "achieved": dom["achieved_GBs"],
                                "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": dom["frac_of_8TBs"], "traffic": traffic,
                                "traffic_source": traffic_src,
                                "algorithmic_bytes_per_launch": comp,
                                # the kernel's own on-chip floors (exact fp32 on the vector ALUs): the 0.80 HBM target
                                # (76 us) lies below both
                                "valu_floor_ms": valu_floor_ms, "lds_floor_ms": lds_floor_ms,
                                # how close the launch is to what bounds THIS formulation: LDS -> VGPR fragment bandwidth
                                "gather_path_ms": path_f, "frac_of_path": comp / path_f / 1e6 / PEAK_HBM_GBS,
                                "ceiling_frac": comp / max(valu_floor_ms, lds_floor_ms) / 1e6 / PEAK_HBM_GBS,
                                "frac_of_lds_floor": lds_floor_ms / dom["gather_tiled_kernel_ms"],
                                "frac_of_valu_floor": valu_floor_ms / dom["gather_tiled_kernel_ms"],
                                "floors_note": "valu: 4.29 G lane-FMAs at the measured 105 lane-FMA/clk/CU (tools/valu_peak.hip) x 256 CUs x "
                                               "2.4 GHz; lds: 17.2 GB of window fragments x 0.78 (anchor sharing) at 256 B/clk/CU",
                                "timing": "HIP event pair around the kernel launch (pips_mixer_input_build_tiled_timed), "
                                          "mean of 10 launches on the real maps"},
            "gather_roofline_bf16": gather_bf16,
            "gather": out}


def config5_leg(device):
    """BASELINE configs[4]: 100-frame clip (synthetic 360x640 frames: demo_images/ is not on the bench box),
    stride 4, N=256 grid at frame 0, visibility-aware chaining (chain_demo.py:40-83) on the cached maps."""
    import torch
    from pips_amd import Pips, drivers
    T, h, w, n = 100, 360, 640, 256
    model = Pips(S=S, stride=4).to(device).eval()
    g = torch.Generator().manual_seed(3)
    rgbs = torch.randint(0, 256, (1, T, 3, h, w), generator=g, dtype=torch.uint8).to(device).float()
    gy, gx = torch.meshgrid(torch.linspace(16, h - 16, 16), torch.linspace(16, w - 16, 16), indexing="ij")
    xy0 = torch.stack([gx.reshape(-1), gy.reshape(-1)], -1).unsqueeze(0).to(device)
    drivers.track_chained(model, rgbs, xy0, iters=ITERS)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    drivers.track_chained(model, rgbs, xy0, iters=ITERS)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"workload": "BASELINE configs[4]: 100 frames 360x640 (synthetic), stride 4, N=256, S=8 windows chained on "
                        "visibility, encoder once per frame", "seconds_per_video": dt, "frames_x_tracks_per_s": T * n / dt}


def torch_rocm_baseline(device):
    """The same forward through stock PyTorch-ROCm ops on this GPU (checker code on device tensors)."""
    import torch
    from oracle import pips_oracle as O
    from pips_amd.weights import init_state_dict
    sd = {k: v.to(device) for k, v in init_state_dict(0).items()}
    xys, rgbs = make_inputs(0, device, 1)
    fn = lambda: O.forward(sd, xys, rgbs, iters=ITERS, stride=STRIDE)
    t_start = time.time()
    with torch.no_grad():
        fn()                                   # MIOpen / rocBLAS warm-up (kernel selection)
        fn()
        torch.cuda.synchronize()
        if time.time() - t_start > 90:
            reps = 1
        else:
            reps = 5
        t = ev_time_ms(fn, reps)
    return {"value": S * NPTS * ITERS / t * 1e3, "unit": "particle-updates/s", "ms_per_step": t,
            "what": f"oracle/pips_oracle.py (bit-identical restatement of nets/pips.py) on cuda tensors, torch {torch.__version__} "
                    "eager fp32 (MIOpen convs, rocBLAS matmuls), same workload as the headline"}


def cpu_baseline():
    """The CPU oracle (port of nets/pips.py forward) on this host: same workload, bounded sample."""
    import torch
    from oracle import pips_oracle as O
    from pips_amd.weights import init_state_dict
    from oracle.hostinfo import effective_cpus
    cores = effective_cpus()                      # cgroup quota, not the 256 logical CPUs
    torch.set_num_threads(cores)
    sd = init_state_dict(0)
    g = torch.Generator().manual_seed(1)
    rgbs = torch.randint(0, 256, (1, S, 3, H, W), generator=g).float()
    xys = torch.rand(1, NPTS, 2, generator=g) * torch.tensor([W - 1.0, H - 1.0])
    O.forward(sd, xys, rgbs, iters=ITERS, stride=STRIDE)            # warm-up
    ts = []
    t_end = time.time() + 20.0
    while len(ts) < 3 or (time.time() < t_end and len(ts) < 8):
        t0 = time.time()
        O.forward(sd, xys, rgbs, iters=ITERS, stride=STRIDE)
        ts.append(time.time() - t0)
    med = statistics.median(ts)
    return {"value": S * NPTS * ITERS / med, "unit": "particle-updates/s", "cores": cores, "kind": "port",
            "kind_note": "port of nets/pips.py's forward with bit-identical outputs (oracle/check_against_reference.py: 0.0); it skips "
                         "the dead per-iteration `fcp` score-map upsample (nets/pips.py:504-511, ~10 % of the reference's CPU time), "
                         "so the unmodified reference would read ~10 % lower; /root/reference does not exist on the bench box",
            "sample": f"{len(ts)} forwards of the same workload (B=1,S=8,368x496,N=256,I=6 fp32), median {med:.3f} s, "
                      f"torch {torch.__version__} CPU ops, oracle/pips_oracle.py"}


class _CpuStandIn:
    """PIPS_BENCH_FAKE=1 (launcher self-test on a box without GPUs): stands in for Pips on CPU tensors."""

    def __call__(self, xys, rgbs, iters=6, **kw):
        b, n, _ = xys.shape
        base = xys.reshape(b, 1, n, 2).repeat(1, rgbs.shape[1], 1, 1) + rgbs.mean(dim=(2, 3, 4)).reshape(b, -1, 1, 1)
        preds = [base + i for i in range(iters)]
        return preds, [base, base] + preds + [preds[-1]] * 2, base.sum(-1), None

    def encode(self, rgbs):                      # (the encoder / tracker split of the particle-sharded leg)
        return rgbs

    def track(self, cache, xys, iters=6, **kw):
        return self(xys, cache, iters=iters)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-stage-profile", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the config 3/4/5 legs and the torch-ROCm baseline")
    ap.add_argument("--config", type=int, default=2, choices=(2, 3, 4),
                    help="2 (default, the headline: B=1/GPU fp32), 3 (B=8/GPU, bf16 MFMA operands) or 4 (BASELINE configs[3]: "
                         "B=4 720x1280 N=4096 fp32 -- with --gpus N the PARTICLES are sharded over the ranks on replicated maps, "
                         "strong scaling, SURVEY 8(e) secondary axis)")
    ap.add_argument("--encode", default="frames", choices=("frames", "replicate"),
                    help="--config 4 with --gpus N: how the ranks get the clips' maps -- 'frames' (default): each rank encodes S/N frames of "
                         "every clip and one all-gather per pyramid level rebuilds the cache (the encoder scales with the ranks too); "
                         "'replicate': every rank encodes all frames (no exchange, Amdahl-capped by the encoder)")
    ap.add_argument("--leg", default=None, choices=("config3", "config4", "config5", "torch_rocm_baseline"),
                    help="run ONE of the extra legs alone and print its JSON (what the rocprofv3 passes under profiles/ wrap)")
    ap.add_argument("--matmul", default="exact", choices=("exact", "split"),
                    help="config 2 only: exact-fp32 MFMA (default, the headline) or the fp32-grade split-bf16 path")
    ap.add_argument("--mixer-stream", default=None, choices=("f32", "bf16"),
                    help="tuning only: residual-stream type of the bf16 mixer in the config-3 legs (default: the module's own)")
    ap.add_argument("--lib", default=None, help="tuning only: bind pips_amd to this build of the library (tools/ab_c3.sh); the "
                                                "driver's runs never pass it")
    args = ap.parse_args(argv)
    gpus = max(1, args.gpus)
    global MIXER_STREAM
    MIXER_STREAM = args.mixer_stream
    if args.lib:
        from pips_amd import _lib as _pl
        _pl.use_library(args.lib)
    if args.leg:
        import torch
        device = torch.device("cuda", 0)
        torch.cuda.set_device(device)
        res = {"config3": config3_leg, "config4": config4_leg, "config5": config5_leg,
               "torch_rocm_baseline": torch_rocm_baseline}[args.leg](device)
        print(json.dumps({args.leg: res}), flush=True)
        return res
    if "WORLD_SIZE" not in os.environ and gpus > 1:
        sys.exit(respawn_under_launcher(gpus))

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
                         f"(python -m torch.distributed.run --nproc-per-node {gpus} bench.py --gpus {gpus} ...)")
    fake = os.environ.get("PIPS_BENCH_FAKE") == "1"
    # one rank per GPU over RCCL; PIPS_BENCH_BACKEND=gloo (launcher smoke test on a 1-GPU or GPU-less box)
    # lets several ranks share a device
    backend = os.environ.get("PIPS_BENCH_BACKEND", "nccl")
    if fake:
        device = torch.device("cpu")
    else:
        if backend != "nccl":
            local = local % torch.cuda.device_count()
        elif local >= torch.cuda.device_count():
            raise SystemExit(f"bench.py: rank {rank} needs cuda:{local}, only {torch.cuda.device_count()} device(s) visible")
        device = torch.device("cuda", local)
        torch.cuda.set_device(device)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl" and not fake:
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group("gloo")
    sync = (lambda: None) if fake else torch.cuda.synchronize

    from pips_amd import dist as pdist
    b_per_gpu = 8 if args.config == 3 else (4 if args.config == 4 else 1)
    c4_n = 64 if fake else 4096
    if fake:
        model = _CpuStandIn()
        xys, rgbs = make_inputs(0 if args.config == 4 else rank, device, b_per_gpu, 32, 32, c4_n if args.config == 4 else 16)
    elif args.config == 4:
        # BASELINE configs[3]: every rank holds the SAME 4 clips (B < G: nothing to shard on the clip axis) and the 64x64 grid
        from pips_amd import Pips
        model = Pips(S=S, stride=STRIDE).to(device).eval()
        model.matmul = args.matmul
        g4 = torch.Generator().manual_seed(1)
        rgbs = torch.randint(0, 256, (4, S, 3, 720, 1280), generator=g4, dtype=torch.uint8).to(device).float()
        gy, gx = torch.meshgrid(torch.linspace(8, 720 - 8, 64), torch.linspace(8, 1280 - 8, 64), indexing="ij")
        xys = torch.stack([gx.reshape(-1), gy.reshape(-1)], -1).unsqueeze(0).repeat(4, 1, 1).to(device)
    else:
        from pips_amd import Pips
        model = Pips(S=S, stride=STRIDE).to(device).eval()               # seeded random init (seed 0)
        if args.config == 3:
            model.mixer_dtype = model.encoder_dtype = torch.bfloat16     # BASELINE configs[2]
            if MIXER_STREAM is not None:
                model.mixer_stream_dtype = torch.bfloat16 if MIXER_STREAM == "bf16" else torch.float32
        model.matmul = args.matmul
        xys, rgbs = make_inputs(rank, device, b_per_gpu)

    # frame-sharded maps need S divisible by the ranks; a single rank has nothing to shard
    enc4 = args.encode if (world > 1 and S % world == 0 and not fake) else "replicate"       # (the CPU stand-in has no pyramid to exchange)

    def step():
        if args.config == 4:          # particles sharded over the ranks on replicated maps + one gather on the particle axis
            return pdist.track_sharded_particles(model, xys, rgbs, iters=ITERS, encode=enc4)
        preds, _, vis, _ = model(xys, rgbs, iters=ITERS)
        if world > 1:
            pdist.all_gather_result(preds[-1], vis)
        return preds

    on_dev = backend == "nccl" and not fake

    def timed_steps(nsteps):
        """The contract's bracket (barrier + synchronize, wall clock over exactly nsteps steps, max over ranks) and, beside
        it, one HIP event per step boundary on the launch stream -> per-step durations (median reported)."""
        if world > 1:
            dist.barrier()
        sync()
        evs = [] if fake else [torch.cuda.Event(enable_timing=True) for _ in range(nsteps + 1)]
        marks = []
        t0 = time.perf_counter()
        for i in range(nsteps):
            if fake:
                marks.append(time.perf_counter())
            else:
                evs[i].record()
            step()
        if fake:
            marks.append(time.perf_counter())
        else:
            evs[nsteps].record()
        sync()
        if world > 1:
            dist.barrier()
        dt = _max_over_ranks(time.perf_counter() - t0, world, device, on_dev)
        per = ([(b - a) * 1e3 for a, b in zip(marks[:-1], marks[1:])] if fake else
               [evs[i].elapsed_time(evs[i + 1]) for i in range(nsteps)])
        return dt, per

    for _ in range(args.warmup):
        step()
    dt, per_step = timed_steps(args.steps)

    npts = 16 if fake else NPTS
    updates = world * b_per_gpu * S * npts * ITERS * args.steps
    if args.config == 4:              # the whole job is the 4 clips x N particles, whatever the number of ranks (strong scaling)
        updates = b_per_gpu * S * c4_n * ITERS * args.steps
    res = {
        "metric": ("particle-updates/sec (B*S*N*iters/s) at S=8 N=4096 720x1280" if args.config == 4 else
                   "particle-updates/sec (B*S*N*iters/s) at S=8 N=256 368x496"),
        "value": updates / dt,
        "unit": "particle-updates/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "ms_per_step_median": statistics.median(per_step),
        "ms_per_step_timing": "ms_per_step: wall clock over all timed steps (barrier + synchronize on both sides, max over ranks); "
                              "ms_per_step_median: median of the per-step gaps between HIP events recorded on the launch stream "
                              "(rank 0)",
        "higher_is_better": True,
        "scaling": "strong" if args.config == 4 else "weak",
        "vs_baseline": None,
        "dtype": ("f32" if args.matmul == "exact" else
                  "f32-grade: split-bf16 (3 exact bf16 terms per fp32 operand, 6 bf16 MFMA products, fp32 accumulate)")
        if args.config in (2, 4) else "bf16 MFMA operands, fp32 accumulate/state",
        "data": "synthetic (uniform 0..255 frames, uniform in-bounds queries, seeded random-init weights)",
        "config": {"workload": ("BASELINE configs[1]: B=1/GPU S=8 368x496 N=256 I=6 fp32 stride 8, encoder included, "
                                "inputs resident in HBM") if args.config == 2 else
                               ("BASELINE configs[2]: B=8/GPU S=8 368x496 N=256 I=6 bf16 operands stride 8, encoder "
                                "included, inputs resident in HBM") if args.config == 3 else
                               ("BASELINE configs[3]: B=4 S=8 720x1280 N=4096 (64x64 grid) I=6 fp32 stride 8, inputs resident in "
                                "HBM; " + ("every rank encodes S/G frames of the 4 clips, one all-gather per pyramid level rebuilds the maps"
                                           if enc4 == "frames" else "every rank encodes the 4 clips (replicated maps)") +
                                " and tracks N/G particles"),
                   "encode": enc4 if args.config == 4 else None,
                   "clips_per_gpu": b_per_gpu,
                   "parallelism": f"particle-sharded x{world} (pips_amd.dist.track_sharded_particles)" if args.config == 4
                   else f"clip-sharded x{world}",
                   "collective": ("none" if world == 1 else f"one all_gather of [x,y,vis] per step"
                                  f"{' on the particle axis' if args.config == 4 else ''}, backend "
                                  f"{'gloo' if (fake or backend != 'nccl') else 'nccl (RCCL)'}")},
    }
    if fake:
        res["data"] = "PIPS_BENCH_FAKE=1: CPU stand-in model, launcher/collective self-test only -- not a measurement"
    main_cfg = args.config == 2 and args.matmul == "exact"
    headline = rank == 0 and world == 1 and main_cfg and not fake
    if headline and not args.no_stage_profile:
        # the same workload on the fp32-grade split-bf16 matrix path (Pips.matmul = "split"; passes the
        # same fp32 parity gates, tests/test_forward_gpu.py) -- reported beside the exact-fp32 headline
        model.matmul = "split"
        for _ in range(args.warmup):
            step()
        dts, per_s = timed_steps(args.steps)
        model.matmul = "exact"
        res["split_bf16"] = {"value": updates / dts, "unit": "particle-updates/s", "ms_per_step": dts / args.steps * 1e3,
                             "ms_per_step_median": statistics.median(per_s),
                             "note": "Pips.matmul='split' (PIPS_FLAG_SPLIT_BF16): mixer GEMMs + the larger convs as "
                                     "6 exact bf16 MFMA products per fp32 product; same parity gates as fp32"}
        stages, kern, gather, split = stage_profile(model, xys, rgbs, device, b_per_gpu)
        res["split_bf16"].update(split)
        res["roofline"] = roofline_object(kern)
        res["gather"] = gather
        res["stages_ms"] = stages
        flop_per_update = 72.2e6
        res["forward_mfma_frac"] = res["value"] / world * flop_per_update / (PEAK_F32_MFMA_TF * 1e12)
    if world > 1 and main_cfg and not args.no_extras:
        # A multi-rank line still describes the workload it scales: the dominant kernel's roofline (rank 0, in situ), the
        # exchange alone, and BASELINE configs[2] -- the multi-GPU config -- weak (8 clips per GPU) and strong (64 clips).
        if not fake and rank == 0:
            res["roofline"] = roofline_object(gemm_roofline(model, device, b_per_gpu))
        dist.barrier()
        res["collective_ms"] = collective_leg(device, world, fake, npts)
        try:
            res["config3"] = config3_leg(device, world, rank, fake, on_dev)
        except Exception as e:                                       # (every rank raises alike: shapes and code are identical)
            res["config3"] = {"error": f"{type(e).__name__}: {e}"}
    if headline and not args.no_extras:
        for name, leg in (("config3", config3_leg), ("config4", config4_leg), ("config5", config5_leg),
                          ("torch_rocm_baseline", torch_rocm_baseline)):
            try:
                torch.cuda.empty_cache()
                res[name] = leg(device)
            except Exception as e:                                   # a leg must not take the headline down
                res[name] = {"error": f"{type(e).__name__}: {e}"}
        torch.cuda.empty_cache()
    if headline and not args.no_cpu_baseline:
        res["cpu_baseline"] = cpu_baseline()
    if rank == 0:
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()
    return res


if __name__ == "__main__":
    main()
