"""A/B of the token-mix output stores' cache policy (variant builds of track.hip with -DPIPS_TM_STORE=k, PIPS_LIB_PATH selects one):
200 bf16 mixer passes at BASELINE configs[2]'s M = 16384 rows, bf16 residual stream; run under rocprofv3 --kernel-trace --stats for the
per-kernel averages, prints the pass time by HIP events."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import _tunelib  # noqa: F401
from pips_amd import ops
from pips_amd.weights import init_state_dict
dev = "cuda:0"
M = int(os.environ.get("PIPS_AB_M", "16384"))
BF16 = os.environ.get("PIPS_AB_BF16", "1") == "1"     # 0: the exact-fp32 mixer (headline: PIPS_AB_M=2048)
arena = ops.pack_weights(init_state_dict(0), torch.device(dev), sections=ops.PACK_FP32 | ops.PACK_BF16)
X = torch.randn(M, 544, generator=torch.Generator().manual_seed(0)).to(dev)
for _ in range(50):
    out = ops.mixer_fwd(arena, X, bf16=BF16, stream_bf16=BF16)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200):
    out = ops.mixer_fwd(arena, X, bf16=BF16, stream_bf16=BF16)
e1.record()
torch.cuda.synchronize()
print("%s: %.4f ms per mixer pass; checksum %.6f" % (os.environ.get("PIPS_LIB_PATH", "product"), e0.elapsed_time(e1) / 200, float(out.double().abs().sum())))
