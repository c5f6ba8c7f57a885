"""Encoder alone (Pips.encode) for F frames of HxW, fp32 or bf16 mode: milliseconds per pass, HIP events.
usage: python tools/encode_bench.py F H W [bf16|split]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import _tunelib  # noqa: F401
from pips_amd import Pips
F, H, W = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
mode = sys.argv[4] if len(sys.argv) > 4 else "f32"
dev = torch.device("cuda:0")
m = Pips(stride=8).to(dev).eval()
if mode == "bf16":
    m.encoder_dtype = m.mixer_dtype = torch.bfloat16
if mode == "split":
    m.matmul = "split"
rgbs = torch.randint(0, 256, (1, F, 3, H, W), generator=torch.Generator().manual_seed(1)).float().to(dev)
for _ in range(2):
    m.encode(rgbs, frames_per_pass=F)
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
reps = 10 if F * H * W < 40e6 else 4
e0.record()
for _ in range(reps):
    m.encode(rgbs, frames_per_pass=F)
e1.record(); e1.synchronize()
print(f"encode F={F} {H}x{W} {mode}: {e0.elapsed_time(e1) / reps:.3f} ms", flush=True)
