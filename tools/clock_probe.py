"""Shader clock and power under a sustained kernel: loops one workload for a few seconds while rocm-smi is sampled.
usage: python tools/clock_probe.py gemm16384 | gemm2048 | conv64 | mixerbf16_16384 | idle"""
import os, sys, subprocess, threading, time, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import _tunelib  # noqa: F401
from pips_amd import ops
dev = "cuda:0"
what = sys.argv[1] if len(sys.argv) > 1 else "gemm16384"
g = torch.Generator().manual_seed(0)
if what.startswith("gemm"):
    M = int(what[4:])
    A = torch.randn(M, 512, generator=g).to(dev)
    W = (torch.randn(2048, 512, generator=g) / math.sqrt(512)).to(dev)
    b = torch.randn(2048, generator=g).to(dev)
    fn = lambda: ops.gemm(A, W, b, 1)
    flops = 2.0 * M * 2048 * 512
elif what == "conv64":
    x = torch.randn(8, 184, 248, 64, generator=g).to(dev)
    w = (torch.randn(64, 3, 3, 64, generator=g) / 24.0).to(dev)
    b = torch.randn(64, generator=g).to(dev)
    fn = lambda: ops.conv_nhwc(x, w, b, 3, 1, 1, want_stats=True)
    flops = 2.0 * 8 * 184 * 248 * 64 * 576
elif what.startswith("mixerbf16_"):
    # the bf16 mixer pass of BASELINE configs[2] (M rows; 12 x {token mix, up-projection + GELU, down-projection + residual})
    from pips_amd.weights import init_state_dict
    M = int(what.split("_")[1])
    arena = ops.pack_weights(init_state_dict(0), torch.device(dev), sections=ops.PACK_FP32 | ops.PACK_BF16)
    X = torch.randn(M, 544, generator=g).to(dev)
    fn = lambda: ops.mixer_fwd(arena, X, bf16=True, stream_bf16=True)
    flops = 2.0 * M * (544 * 512 + 12 * (2 * 512 * 2048 + 2 * 8 * 32 * 512 / 8) + 512 * 130)
else:
    fn, flops = None, 0.0
samples = []


def sampler():
    for _ in range(6):
        time.sleep(0.4)
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout
        keep = [ln.strip() for ln in out.splitlines() if "sclk" in ln or "Power" in ln or "mclk" in ln]
        samples.append(" | ".join(keep))


th = threading.Thread(target=sampler)
th.start()
n, t0 = 0, time.time()
if fn is not None:
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.time() - t0 < 3.0:
        for _ in range(50):
            fn()
        n += 50
        torch.cuda.synchronize()
    e1.record(); e1.synchronize()
    print(f"{what}: {n} launches, {e0.elapsed_time(e1) / n * 1e3:.1f} us each, {flops * n / e0.elapsed_time(e1) / 1e9:.1f} TF sustained")
th.join()
for s in samples:
    print("   ", s)
