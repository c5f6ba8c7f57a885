"""BASELINE configs[2] share of one GPU (8 clips, bf16) as ONE forward of 8 clips against TWO concurrent forwards of 4 clips on two
streams (one module, per-stream scratch): does the second stream fill the first one's idle tails?"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import _tunelib  # noqa: F401
from pips_amd import Pips
dev = torch.device("cuda:0")
m = Pips(stride=8).to(dev).eval()
m.mixer_dtype = m.encoder_dtype = torch.bfloat16
g = torch.Generator().manual_seed(1)
rgbs = torch.randint(0, 256, (8, 8, 3, 368, 496), generator=g).float().to(dev)
xys = (torch.rand(8, 256, 2, generator=g) * torch.tensor([495.0, 367.0])).to(dev)


def one(reps):
    for _ in range(reps):
        m(xys, rgbs, iters=6)


def split(reps, k):
    sts = [torch.cuda.Stream() for _ in range(k)]
    per = 8 // k
    for _ in range(reps):
        for i, st in enumerate(sts):
            with torch.cuda.stream(st):
                m(xys[i * per:(i + 1) * per], rgbs[i * per:(i + 1) * per], iters=6)
    for st in sts:
        st.synchronize()


def timed(fn, *a):
    fn(2, *a) if a else fn(2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn(10, *a) if a else fn(10)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 10 * 1e3


for rnd in range(2):
    print(f"round {rnd}: one forward of 8 clips {timed(one):.2f} ms | 2 streams x 4 clips {timed(split, 2):.2f} ms | 4 streams x 2 clips {timed(split, 4):.2f} ms", flush=True)
