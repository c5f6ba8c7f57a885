"""Time the forward and the gather kernel at an arbitrary config (e.g. BASELINE config 4).
usage: python tools/config_probe.py B H W N [grid|rand] [iters]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import _tunelib  # noqa: F401  (PIPS_LIB_PATH -> pips_amd._lib.use_library)
from pips_amd import Pips, ops
B, H, W, N = map(int, sys.argv[1:5])
mode = sys.argv[5] if len(sys.argv) > 5 else "rand"
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 6
S, stride = 8, 8
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
rgbs = torch.randint(0, 256, (B, S, 3, H, W), generator=g, dtype=torch.uint8).to(dev).float()
if mode == "grid":
    n = int(round(N ** 0.5))
    gy, gx = torch.meshgrid(torch.linspace(8, H - 8, n), torch.linspace(8, W - 8, n), indexing="ij")
    xys = torch.stack([gx.reshape(-1), gy.reshape(-1)], -1).unsqueeze(0).repeat(B, 1, 1)
else:
    xys = torch.rand(B, N, 2, generator=g) * torch.tensor([W - 1.0, H - 1.0])
xys = xys.to(dev)
m = Pips(stride=stride).to(dev).eval()
m.matmul = os.environ.get("PIPS_MATMUL", "exact")       # exact | split
def ev(fn, reps):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps
t = ev(lambda: m(xys, rgbs, iters=iters), 3)
print(f"[{m.matmul}] B={B} {H}x{W} N={N} {mode} I={iters}: forward {t:.2f} ms -> {B*S*N*iters/t*1e3:.3e} particle-updates/s", flush=True)
arena = m._packed(dev)
H8, W8 = H // stride, W // stride
F = B * S
t_enc = ev(lambda: ops.encoder_fwd(arena, rgbs.reshape(F, 3, H, W), stride), 2)
pyr = ops.encoder_fwd(arena, rgbs.reshape(F, 3, H, W), stride)
M = B * N * S
coords = (xys / stride).reshape(B, N, 1, 2).repeat(1, 1, S, 1).reshape(M, 2).contiguous()
ffeats = torch.randn(M, 128, device=dev)
for jitter in (0.0, 2.0):
    c = coords + torch.randn(M, 2, device=dev) * jitter
    tg = ev(lambda: ops.mixer_input_build(pyr, B, H8, W8, ffeats, c), 5)
    tt_ = ev(lambda: ops.mixer_input_build_tiled(pyr, B, H8, W8, ffeats, c), 5)
    print(f"  tiled gather (jitter {jitter}): {tt_*1e3:.1f} us   direct: {tg*1e3:.1f} us", flush=True)
    lv = sum((H8 >> l) * (W8 >> l) for l in range(4))
    comp = F * lv * 512 + M * (512 + 8) + M * 544 * 4
    gath = M * (4 * 64 * 512 + 512 + 8 + 544 * 4)
    print(f"  gather (coord jitter {jitter} px): {tg*1e3:.1f} us  compulsory {comp/1e6:.1f} MB -> {comp/tg/1e6:.0f} GB/s "
          f"({comp/tg/1e6/8000*100:.1f}% of 8 TB/s), L2-level {gath/tg/1e6:.0f} GB/s", flush=True)
X = ops.mixer_input_build(pyr, B, H8, W8, ffeats, coords)
t_mix = ev(lambda: ops.mixer_fwd(arena, X), 2)
print(f"  encoder {t_enc:.2f} ms, mixer pass {t_mix:.2f} ms ({2*207.1e6*M/8/t_mix/1e9:.1f} TF)")
