"""bf16-operand channel-mix GEMMs at M rows (default 16384 = config 3): in-situ HIP-event time per launch for the tile the
PIPS_BF16_BIG hook selects, next to torch.matmul in bf16 (hipBLASLt) on the same shapes -- a yardstick, not product code."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import _tunelib  # noqa: F401  (PIPS_LIB_PATH -> pips_amd._lib.use_library)
from pips_amd import ops
from pips_amd.weights import init_state_dict
dev = "cuda:0"
M = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
arena = ops.pack_weights(init_state_dict(0), torch.device(dev))
X = torch.randn(M, 544, generator=torch.Generator().manual_seed(0)).to(dev)
best = None
for _ in range(6):
    _, ms = ops.mixer_fwd_timed(arena, X, flags=2)
    best = ms if best is None else {k: min(best[k], ms[k]) for k in ms}
fl = 2.0 * M * 2048 * 512
print(f"BIG={os.environ.get('PIPS_BF16_BIG')} M={M}: up {best['up_proj']*1e3:.1f} us ({fl/best['up_proj']/1e9:.0f} TF)  "
      f"down {best['down_proj']*1e3:.1f} us ({fl/best['down_proj']/1e9:.0f} TF)  in {best['in_proj']*1e3:.1f} us")
if os.environ.get("PIPS_PROBE_BLAS"):
    def t(fn, n=50):
        for _ in range(5): fn()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); e1.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    for (N, K) in ((2048, 512), (512, 2048)):
        a = torch.randn(M, K, device=dev).bfloat16(); w = torch.randn(N, K, device=dev).bfloat16()
        us = t(lambda: torch.matmul(a, w.t()))
        print(f"hipBLASLt bf16 M={M} N={N} K={K}: {us:.1f} us  {2.0*M*N*K/us/1e6:.0f} TF")
