"""In-situ timing of one mixer pass (M = B*N*8 rows) for tile-choice A/B (PIPS_GEMM_TILE)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import _tunelib  # noqa: F401  (PIPS_LIB_PATH -> pips_amd._lib.use_library)
from pips_amd import ops
from pips_amd.weights import init_state_dict
dev = "cuda:0"
M = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
BF16 = len(sys.argv) > 2 and sys.argv[2] == "bf16"
SPLIT = len(sys.argv) > 2 and sys.argv[2] == "x3"
arena = ops.pack_weights(init_state_dict(0), torch.device(dev))
X = torch.randn(M, 544, generator=torch.Generator().manual_seed(0)).to(dev)
for _ in range(3): ops.mixer_fwd(arena, X, bf16=BF16, split=SPLIT)
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): ops.mixer_fwd(arena, X, bf16=BF16, split=SPLIT)
e1.record(); e1.synchronize()
ms = e0.elapsed_time(e1) / 10
print(f"tile={os.environ.get('PIPS_GEMM_TILE')} bf16={BF16} x3={SPLIT} M={M}: mixer pass {ms*1e3:.1f} us  ({2*207.1e6*M/8/ms/1e9:.1f} TF incl. token-mix)")
