import os, sys, time
t0 = time.time()
def lap(msg):
    global t0
    print(f"[{time.time()-t0:7.2f}s] {msg}", flush=True); t0 = time.time()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
lap("import torch")
print("cpu_count", os.cpu_count(), "threads", torch.get_num_threads(), "affinity", len(os.sched_getaffinity(0)), flush=True)
try:
    print(open("/sys/fs/cgroup/cpu.max").read().strip())
except Exception as e:
    print("no cgroup cpu.max", e)
import _tunelib  # noqa: F401  (PIPS_LIB_PATH -> pips_amd._lib.use_library)
from pips_amd import Pips
from pips_amd.weights import init_state_dict
sd = init_state_dict(0, tamed=True); lap("init_state_dict")
m = Pips(); lap("Pips()")
m = m.to("cuda:0").eval(); torch.cuda.synchronize(); lap("to cuda")
g = torch.Generator().manual_seed(1)
rgbs = torch.randint(0, 256, (1, 8, 3, 368, 496), generator=g).float()
xys = torch.rand(1, 256, 2, generator=g) * torch.tensor([495.0, 367.0])
lap("inputs")
xc, rc = xys.cuda(), rgbs.cuda(); torch.cuda.synchronize(); lap("H2D")
out = m(xc, rc, iters=6); torch.cuda.synchronize(); lap("first forward (incl. weight pack)")
for i in range(3):
    out = m(xc, rc, iters=6); torch.cuda.synchronize(); lap(f"forward {i}")
from oracle import pips_oracle as O
for nt in (8, 32, 64):
    torch.set_num_threads(nt)
    O.forward(sd, xys, rgbs, iters=1, stride=8); lap(f"oracle warm nt={nt} iters=1")
    O.forward(sd, xys, rgbs, iters=6, stride=8); lap(f"oracle nt={nt} iters=6")
