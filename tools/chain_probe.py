"""Config-5 style measurement: T-frame video, N particles, stride 4, chained 8-frame windows."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import _tunelib  # noqa: F401  (PIPS_LIB_PATH -> pips_amd._lib.use_library)
from pips_amd import Pips, drivers
T, H, W, N = (int(a) for a in (sys.argv[1:5] if len(sys.argv) > 4 else (100, 360, 640, 256)))
dev = "cuda:0"
g = torch.Generator().manual_seed(0)
base = torch.randint(0, 256, (1, 1, 3, H, W), generator=g).float()
video = torch.cat([(base.roll(3 * t, 4) * (1 - 0.002 * t)).round() for t in range(T)], dim=1).to(dev)
n_ = int(round(N ** 0.5))
gy, gx = torch.meshgrid(torch.linspace(8, H - 8, n_), torch.linspace(8, W - 8, n_), indexing="ij")
xy0 = torch.stack([gx.reshape(-1), gy.reshape(-1)], -1).unsqueeze(0).to(dev)
m = Pips(stride=4).to(dev).eval()
drivers.track_chained(m, video[:, :16], xy0, iters=6)       # warm-up
torch.cuda.synchronize(); t0 = time.time()
cache = m.encode(video); torch.cuda.synchronize(); t1 = time.time()
tr = drivers.track_chained(m, video, xy0, iters=6); torch.cuda.synchronize(); t2 = time.time()
print(f"T={T} {H}x{W} N={xy0.shape[1]} stride 4: encode {1e3*(t1-t0):.1f} ms ({T/(t1-t0):.0f} frames/s); "
      f"chained tracking incl. encode {1e3*(t2-t1):.1f} ms -> {T*xy0.shape[1]/(t2-t1):.0f} frame-tracks/s; finite={bool(torch.isfinite(tr).all())}")
