"""bf16 mixer pass at M rows: token-mix launch + two GEMMs / + fused FeedForward / one launch per layer, same process.
Tuning library (PIPS_LIB_PATH=pips_amd/libpips_hip_tune.so) with PIPS_MIXER_LAYER=0 so that the plain entry point is the
two-GEMM route at every M; the fused forms are forced through their own entry points.  Interleaved rounds."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import _tunelib  # noqa: F401
from pips_amd import ops
from pips_amd.weights import init_state_dict
dev = "cuda:0"
M = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
arena = ops.pack_weights(init_state_dict(0), torch.device(dev))
X = torch.randn(M, 544, generator=torch.Generator().manual_seed(0)).to(dev)


def t(fused, reps=20):
    for _ in range(3):
        ops.mixer_fwd(arena, X, bf16=True, fused=fused)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.mixer_fwd(arena, X, bf16=True, fused=fused)
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps


for rnd in range(3):
    r = {k: t(v) for k, v in (("gemms", False), ("ffn", "ffn"), ("layer", "layer"))}
    print(f"M={M} PIPS_MIXER_LAYER={os.environ.get('PIPS_MIXER_LAYER')} round {rnd}: " +
          "  ".join(f"{k} {v * 1e3:.0f} us/pass ({v * 1e3 / 12:.1f} per layer incl. 1/12 of in-proj+head)" for k, v in r.items()), flush=True)
