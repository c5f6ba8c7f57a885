import sys, os
sys.path.insert(0, os.getcwd())
import torch
import _tunelib  # noqa: F401  (PIPS_LIB_PATH -> pips_amd._lib.use_library)
from pips_amd import ops
from pips_amd.weights import init_state_dict
arena = ops.pack_weights(init_state_dict(0, tamed=True), torch.device("cuda:0"))
X = torch.randn(2048, 544, generator=torch.Generator().manual_seed(0)).cuda()
d0 = ops.mixer_fwd(arena, X); d1 = ops.mixer_fwd(arena, X, split=True); d2 = ops.mixer_fwd(arena, X, bf16=True)
print("delta scale", d0.abs().mean().item(), "x3 vs fp32 max diff", (d1-d0).abs().max().item(), "bf16 vs fp32 max diff", (d2-d0).abs().max().item())
