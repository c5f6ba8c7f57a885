import os, sys, math, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import _tunelib  # noqa: F401  (PIPS_LIB_PATH -> pips_amd._lib.use_library)
from pips_amd import ops
dev = "cuda:0"
F_, H, W = 64, 184, 248
x = torch.randn(F_, H, W, 64, device=dev); w = (torch.randn(64, 3, 3, 64, device=dev) / 24).bfloat16(); b = torch.randn(64, device=dev)
def t(n=10):
    for _ in range(3): ops.conv_nhwc_bf16(x, w, b, 3, 1, 1, want_stats=True)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): ops.conv_nhwc_bf16(x, w, b, 3, 1, 1, want_stats=True)
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
us = t()
print(f"PIPS_CONV_C64={os.environ.get('PIPS_CONV_C64')}: {us:.1f} us  {2.0*F_*H*W*64*576/us/1e6:.0f} TF")
