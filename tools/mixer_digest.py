"""SHA-1 of one mixer pass's output on fixed inputs (bitwise A/B of two builds / tuning settings).  usage: python tools/mixer_digest.py [M]"""
import os, sys, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import _tunelib  # noqa: F401
from pips_amd import ops
from pips_amd.weights import init_state_dict
dev = "cuda:0"
M = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
arena = ops.pack_weights(init_state_dict(0), torch.device(dev))
X = torch.randn(M, 544, generator=torch.Generator().manual_seed(0)).to(dev)
out = ops.mixer_fwd(arena, X)
out = out[0] if isinstance(out, (tuple, list)) else out
torch.cuda.synchronize()
print("M=%d digest %s  absmax %.6f" % (M, hashlib.sha1(out.cpu().numpy().tobytes()).hexdigest()[:16], float(out.abs().max())))
