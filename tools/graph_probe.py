"""Whole-forward time at config 2: eager launches vs a captured hipGraph replay (is launch overhead visible?)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import _tunelib  # noqa: F401  (PIPS_LIB_PATH -> pips_amd._lib.use_library)
from pips_amd import Pips
dev = torch.device("cuda:0")
m = Pips(S=8, stride=8).to(dev).eval()
m.matmul = sys.argv[1] if len(sys.argv) > 1 else "exact"
g = torch.Generator().manual_seed(1)
rgbs = torch.randint(0, 256, (1, 8, 3, 368, 496), generator=g).float().to(dev)
xys = (torch.rand(1, 256, 2, generator=g) * torch.tensor([495.0, 367.0])).to(dev)
def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
eager = timeit(lambda: m(xys, rgbs, iters=6))
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side): m(xys, rgbs, iters=6)
torch.cuda.current_stream().wait_stream(side)
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr): out = m(xys, rgbs, iters=6)
graph = timeit(gr.replay)
print(f"{m.matmul}: eager {eager:.3f} ms   graph replay {graph:.3f} ms")
