"""BASELINE configs[2] on one GPU (8 clips, bf16 mode): ONE forward of 8 clips against TWO forwards of 4 clips on two streams, in flight
together -- do launch floors and kernel tails of one stream fill with the other's work?  (The module keeps its scratch per stream.)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import _tunelib  # noqa: F401
import bench
from pips_amd import Pips
dev = torch.device("cuda:0")
model = Pips(S=8, stride=8).to(dev).eval()
model.mixer_dtype = model.encoder_dtype = torch.bfloat16
xys, rgbs = bench.make_inputs(0, dev, 8)
s = [torch.cuda.Stream() for _ in range(4)]


def one():
    return model(xys, rgbs, iters=6)


def split(k):
    n = 8 // k
    outs = []
    cur = torch.cuda.current_stream()
    for i in range(k):
        s[i].wait_stream(cur)
        with torch.cuda.stream(s[i]):
            outs.append(model(xys[i * n:(i + 1) * n], rgbs[i * n:(i + 1) * n], iters=6))
    for i in range(k):
        cur.wait_stream(s[i])
    return outs


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


ref = one()
sp = split(2)
torch.cuda.synchronize()
err = float((torch.cat([o[0][-1] for o in sp], 0) - ref[0][-1]).abs().max())
for rnd in range(2):
    print("one forward of 8 clips: %.3f ms;  2 x 4 clips on two streams: %.3f ms;  4 x 2 clips on four streams: %.3f ms;  2 x 4 clips one after the other: %.3f ms   (max |d traj| split vs whole %.2e px)"
          % (timed(one), timed(lambda: split(2)), timed(lambda: split(4)), timed(lambda: [model(xys[:4], rgbs[:4], iters=6), model(xys[4:], rgbs[4:], iters=6)]), err))
