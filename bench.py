#!/usr/bin/env python
"""bench.py -- particle-updates/s of the PIPs inference hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--no-cpu-baseline]

One "step" = one whole ``Pips.forward`` (encoder + 6 update iterations) over one batch of
synthetic clips already resident in HBM.  Workload = BASELINE.json configs[1]:
B=1 clip per GPU, S=8 frames, 368x496, N=256 particles, I=6 iterations, fp32, stride 8,
seeded random-init weights (the reference checkpoint is not obtainable offline).
N>1: one process per GPU (torch.distributed.run), clips sharded on the batch axis
(weak scaling), one RCCL all-gather of the final [x,y,vis] per step.

Prints ONE JSON line (rank 0): the driver contract plus
  "roofline"     -- dominant kernel (fp32-MFMA GEMM of the mixer) vs the 157.3 TF fp32 peak,
                    timed live with HIP events on the launch stream;
  "gather"       -- the fused correlation-gather kernel vs the HBM roofline (compulsory bytes);
  "stages_ms"    -- where one forward's time goes;
  "cpu_baseline" -- the CPU oracle (a port of the reference forward) on this host, N=1 only.
"""
from __future__ import annotations

import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch                       # noqa: E402
import torch.distributed as dist   # noqa: E402

B_PER_GPU, S, H, W, NPTS, ITERS, STRIDE = 1, 8, 368, 496, 256, 6, 8
PEAK_F32_MFMA_TF = 157.3           # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_BF16_MFMA_TF = 2500.0         # MI355X_MICROARCH.md: ~2.5 PF dense bf16 MFMA (16x the fp32 MFMA rate)
PEAK_HBM_GBS = 8000.0              # spec; 6290 measured-achievable


def make_inputs(rank, device):
    g = torch.Generator().manual_seed(1 + rank)
    rgbs = torch.randint(0, 256, (B_PER_GPU, S, 3, H, W), generator=g).float()
    xys = torch.rand(B_PER_GPU, NPTS, 2, generator=g) * torch.tensor([W - 1.0, H - 1.0])
    return xys.to(device), rgbs.to(device)


def ev_time_ms(fn, reps):
    """Average milliseconds of fn() over reps calls, HIP events on the current stream."""
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps


def stage_profile(model, xys, rgbs, device):
    """Per-stage and per-kernel timings with the staged C-ABI entry points (same kernels)."""
    from pips_amd import ops, _lib
    from pips_amd.weights import MIX_DEPTH
    lib = _lib.load()
    arena = model._packed(device)
    F = B_PER_GPU * S
    H8, W8 = H // STRIDE, W // STRIDE
    M = B_PER_GPU * NPTS * S
    frames = rgbs.reshape(F, 3, H, W)
    pyr = ops.encoder_fwd(arena, frames, STRIDE)
    out = {}
    out["encoder"] = min(ev_time_ms(lambda: ops.encoder_fwd(arena, frames, STRIDE), 3) for _ in range(3))
    g = torch.Generator().manual_seed(0)
    ffeats = torch.randn(M, 128, generator=g).to(device)
    coords = (torch.rand(M, 2, generator=g) * torch.tensor([W8 - 1.0, H8 - 1.0])).to(device)
    X = ops.mixer_input_build(pyr, B_PER_GPU, H8, W8, ffeats, coords)
    t_gather = ev_time_ms(lambda: ops.mixer_input_build(pyr, B_PER_GPU, H8, W8, ffeats, coords), 20)
    out["mixer_input(gather)"] = t_gather * ITERS
    delta = ops.mixer_fwd(arena, X)
    out["mixer"] = ev_time_ms(lambda: ops.mixer_fwd(arena, X), 5) * ITERS
    c0 = coords.clone()
    out["state_update"] = ev_time_ms(
        lambda: ops.state_update(arena, delta, ffeats, coords, c0, B_PER_GPU, NPTS, float(STRIDE)), 10) * ITERS

    # dominant kernel: the channel-mix GEMMs (igemm_f32_kernel), timed IN SITU: a real mixer pass
    # on the real weights/activations with a HIP event pair around every GEMM launch on the
    # launch stream (pips_mixer_fwd_timed); mean over the 12 layers and 5 passes.
    ups, downs, ovh = [], [], []
    for _ in range(5):
        _, t = ops.mixer_fwd_timed(arena, X)
        ups.append(t["up_proj"])
        downs.append(t["down_proj"])
        ovh.append(t["event_overhead"])
    # an event pair costs a marker-to-marker gap even with nothing between; subtract it
    t_ovh = sum(ovh) / len(ovh)
    t_up = sum(ups) / len(ups) - t_ovh
    t_down = sum(downs) / len(downs) - t_ovh
    flops = 2.0 * M * 2048 * 512
    kern = {
        "up_proj(M=%d,N=2048,K=512)" % M: {"ms": t_up, "tflops": flops / t_up / 1e9},
        "down_proj(M=%d,N=512,K=2048)" % M: {"ms": t_down, "tflops": flops / t_down / 1e9},
    }
    # the same two launches on the split-bf16 path (gemm_x3_kernel): fp32-equivalent rate, and the bf16
    # MFMA work actually issued (6 products per fp32 product) against the dense bf16 peak
    ups, downs, ovh = [], [], []
    for _ in range(5):
        _, t = ops.mixer_fwd_timed(arena, X, flags=16)
        ups.append(t["up_proj"])
        downs.append(t["down_proj"])
        ovh.append(t["event_overhead"])
    s_ovh = sum(ovh) / len(ovh)
    s_up, s_down = sum(ups) / len(ups) - s_ovh, sum(downs) / len(downs) - s_ovh
    split = {
        "up_proj_ms": s_up, "down_proj_ms": s_down,
        "fp32_equiv_tflops": {"up_proj": flops / s_up / 1e9, "down_proj": flops / s_down / 1e9},
        "bf16_mfma_tflops_issued": 6 * flops / max(s_up, s_down) / 1e9,
        "frac_of_bf16_peak": 6 * flops / max(s_up, s_down) / 1e9 / PEAK_BF16_MFMA_TF,
        "mixer_ms": ev_time_ms(lambda: ops.mixer_fwd(arena, X, split=True), 5) * ITERS,
        "encoder_ms": min(ev_time_ms(lambda: ops.encoder_fwd(arena, frames, STRIDE, split=True), 3) for _ in range(3)),
    }
    # gather: compulsory bytes per launch: pyramid + ffeats + coords + fcorrs
    lv = sum((H8 >> l) * (W8 >> l) for l in range(4))
    comp_bytes = B_PER_GPU * S * (lv * 128 * 4 + NPTS * 128 * 4 + NPTS * 8 + NPTS * 196 * 4)
    gathered_bytes = M * (4 * 64 * 128 * 4 + 128 * 4 + 8 + 196 * 4)
    gather = {
        "kernel": "mixer_input_kernel",
        "ms_per_launch": t_gather,
        "compulsory_GBs": comp_bytes / t_gather / 1e6,
        "gathered_GBs(L2-level)": gathered_bytes / t_gather / 1e6,
        "frac_of_hbm_peak": comp_bytes / t_gather / 1e6 / PEAK_HBM_GBS,
        "note": "config 2's 18 MB footprint is L2/MALL-resident; HBM fraction is meaningful at config 4",
    }
    return out, kern, gather, split


def cpu_baseline():
    """The CPU oracle (port of nets/pips.py forward) on this host: same workload, bounded sample."""
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
            "sample": f"{len(ts)} forwards of the same workload (B=1,S=8,368x496,N=256,I=6 fp32), median {med:.3f} s, "
                      f"torch {torch.__version__} CPU ops, oracle/pips_oracle.py"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-stage-profile", action="store_true")
    ap.add_argument("--config", type=int, default=2, choices=(2, 3),
                    help="2 (default, the headline: B=1/GPU fp32) or 3 (B=8/GPU, bf16 MFMA operands)")
    ap.add_argument("--matmul", default="exact", choices=("exact", "split"),
                    help="config 2 only: exact-fp32 MFMA (default, the headline) or the fp32-grade split-bf16 path")
    args = ap.parse_args()
    global B_PER_GPU
    if args.config == 3:
        B_PER_GPU = 8

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # one rank per GPU; PIPS_BENCH_BACKEND=gloo (smoke-testing the launcher on a 1-GPU box) lets
    # several ranks share a device
    backend = os.environ.get("PIPS_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local = local % torch.cuda.device_count()
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
    assert world == max(1, args.gpus) or world == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    from pips_amd import Pips, dist as pdist
    model = Pips(S=S, stride=STRIDE).to(device).eval()               # seeded random init (seed 0)
    if args.config == 3:
        model.mixer_dtype = model.encoder_dtype = torch.bfloat16     # BASELINE configs[2]
    model.matmul = args.matmul
    xys, rgbs = make_inputs(rank, device)

    def step():
        preds, _, vis, _ = model(xys, rgbs, iters=ITERS)
        if world > 1:
            pdist.all_gather_result(preds[-1], vis)
        return preds

    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=device if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    updates = world * B_PER_GPU * S * NPTS * ITERS * args.steps
    res = {
        "metric": "particle-updates/sec (B*S*N*iters/s) at S=8 N=256 368x496",
        "value": updates / dt,
        "unit": "particle-updates/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": ("f32" if args.matmul == "exact" else
                  "f32-grade: split-bf16 (3 exact bf16 terms per fp32 operand, 6 bf16 MFMA products, fp32 accumulate)")
        if args.config == 2 else "bf16 MFMA operands, fp32 accumulate/state",
        "data": "synthetic (uniform 0..255 frames, uniform in-bounds queries, seeded random-init weights)",
        "config": {"workload": ("BASELINE configs[1]: B=1/GPU S=8 368x496 N=256 I=6 fp32 stride 8, encoder included, "
                                "inputs resident in HBM") if args.config == 2 else
                               ("BASELINE configs[2]: B=8/GPU S=8 368x496 N=256 I=6 bf16 operands stride 8, encoder "
                                "included, inputs resident in HBM"),
                   "clips_per_gpu": B_PER_GPU, "parallelism": f"clip-sharded x{world}"},
    }
    if rank == 0 and world == 1 and args.config == 2 and args.matmul == "exact" and not args.no_stage_profile:
        # the same workload on the fp32-grade split-bf16 matrix path (Pips.matmul = "split"; passes the
        # same fp32 parity gates, tests/test_forward_gpu.py) -- reported beside the exact-fp32 headline
        model.matmul = "split"
        for _ in range(args.warmup):
            step()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        dts = time.perf_counter() - t1
        model.matmul = "exact"
        res["split_bf16"] = {"value": updates / dts, "unit": "particle-updates/s", "ms_per_step": dts / args.steps * 1e3,
                             "note": "Pips.matmul='split' (PIPS_FLAG_SPLIT_BF16): mixer GEMMs + the larger convs as "
                                     "6 exact bf16 MFMA products per fp32 product; same parity gates as fp32"}
    if rank == 0 and not args.no_stage_profile and args.config == 2 and args.matmul == "exact":
        stages, kern, gather, split = stage_profile(model, xys, rgbs, device)
        if "split_bf16" in res:
            res["split_bf16"].update(split)
        dom = max(kern.items(), key=lambda kv: kv[1]["ms"])
        # HBM-side bytes per launch of that kernel from the committed PMC passes (separate rocprofv3
        # --pmc FETCH_SIZE / WRITE_SIZE runs of this command; profiles/r1_pmc_traffic.json)
        traffic = None
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r1_pmc_traffic.json")))["kernels"]
            key = "pips::igemm_f32_kernel<64, 64, 2, 2, 1, false>" if dom[0].startswith("up") else \
                "pips::igemm_f32_kernel<64, 64, 2, 2, 2, false>"
            traffic = pmc[key]["hbm_bytes"]
        except (OSError, KeyError, ValueError):
            pass
        # both GEMM shapes run the same kernel template; report the slower (dominant) launch
        res["roofline"] = {"bound": "mfma", "achieved": dom[1]["tflops"], "peak": PEAK_F32_MFMA_TF,
                           "unit": "TFLOP/s", "frac": dom[1]["tflops"] / PEAK_F32_MFMA_TF, "traffic": traffic,
                           "kernel": "igemm_f32_kernel " + dom[0], "launch_ms": dom[1]["ms"],
                           "timing": "HIP event pair around each in-situ launch, empty-pair overhead subtracted",
                           "all": kern}
        res["gather"] = gather
        res["stages_ms"] = stages
        flop_per_update = 72.2e6
        res["forward_mfma_frac"] = res["value"] / world * flop_per_update / (PEAK_F32_MFMA_TF * 1e12)
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.config == 2:
        res["cpu_baseline"] = cpu_baseline()
    if rank == 0:
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
