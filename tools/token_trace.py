"""Phase timeline of token_mix_kernel at the headline's 256 particles (variant build: sh tools/build_variant.sh tt track -DPIPS_TUNING -DPIPS_TOKEN_TRACE;
PIPS_LIB_PATH=build/libpips_tt.so).  Stamps: 0 kernel start, 1 loads requested + weights staged, 2 LayerNorm-1 statistics, 3 token MLP, 4 LayerNorm-2 statistics, 5 stores issued."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import _tunelib  # noqa: F401
from pips_amd import ops, _lib
from pips_amd.weights import init_state_dict
dev = "cuda:0"
M = 2048
arena = ops.pack_weights(init_state_dict(0), torch.device(dev))
X = torch.randn(M, 544, generator=torch.Generator().manual_seed(0)).to(dev)
for _ in range(3):
    ops.mixer_fwd(arena, X)
torch.cuda.synchronize()
lib = _lib.load()
buf = np.zeros(256 * 8, dtype=np.uint64)
rc = lib.pips_debug_token_trace(buf.ctypes.data_as(ctypes.c_void_p))
t = buf.reshape(256, 8).astype(np.int64)
d = np.diff(t[:, :6], axis=1)
names = ["request + stage weights", "LayerNorm-1 statistics (incl. the load's round trip)", "LN-1 apply + token MLP", "LayerNorm-2 statistics", "LN-2 apply + stores issued"]
print("rc", rc, "-- last token-mix launch of a mixer pass, 256 blocks; shader clocks (2.39 GHz), median over blocks")
for k, nm in enumerate(names):
    print("  %-52s %7.0f clk = %5.2f us" % (nm, np.median(d[:, k]), np.median(d[:, k]) / 2390.0))
print("  %-52s %7.0f clk = %5.2f us" % ("start -> stores issued", np.median(t[:, 5] - t[:, 0]), np.median(t[:, 5] - t[:, 0]) / 2390.0))
print("  first block start -> last block end %.2f us" % ((t[:, 5].max() - t[:, 0].min()) / 2390.0))
