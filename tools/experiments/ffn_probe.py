"""The bf16 mixer pass (12 layers) at M rows, per layer: how the fused FeedForward scales with the number of 64-row blocks.
   python tools/ffn_probe.py (the fused route, forced);  PIPS_FFN_FUSED=0 python tools/ffn_probe.py (the two-GEMM route)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import _tunelib  # noqa: F401  (PIPS_LIB_PATH -> pips_amd._lib.use_library)
from pips_amd import ops
from pips_amd.weights import init_state_dict
dev = torch.device("cuda:0")
arena = ops.pack_weights(init_state_dict(0), dev)
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for M in [int(m) for m in os.environ.get("FFN_MS", "4096,8192,16384,32768,65536").split(",")]:
    X = torch.randn(M, 544, generator=torch.Generator().manual_seed(0)).to(dev)
    us = t(lambda: ops.mixer_fwd(arena, X, bf16=True, fused=os.environ.get('PIPS_FFN_FUSED', '1') == '1'))
    print(f"M={M:6d} ({M//64:4d} blocks): mixer pass {us:8.1f} us = {us/12:6.1f} us per layer (token-mix + FeedForward)   FFN_FUSED={os.environ.get('PIPS_FFN_FUSED','1')}")
