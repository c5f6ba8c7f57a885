"""Phase timeline of token_mix_mfma_kernel<true> at BASELINE configs[2]'s 2048 particles, bf16 residual stream (variant build:
sh tools/build_variant.sh tt track -DPIPS_TOKEN_TRACE; PIPS_LIB_PATH=build/libpips_tt.so).  Stamps of wave 0 of blocks 0..255: 0 kernel start,
1 tile requested + weights / biases loaded, 2 LayerNorm-1 statistics (incl. the tile's round trip), 3 the 16 channel slots (LN-1 apply, 3 MFMAs,
GELU, residual, stream stores), 4 LayerNorm-2 statistics, 5 LN-2 apply + stores issued."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import _tunelib  # noqa: F401
from pips_amd import ops, _lib
from pips_amd.weights import init_state_dict
dev = "cuda:0"
M = 16384
arena = ops.pack_weights(init_state_dict(0), torch.device(dev), sections=ops.PACK_FP32 | ops.PACK_BF16)
X = torch.randn(M, 544, generator=torch.Generator().manual_seed(0)).to(dev)
for _ in range(3):
    ops.mixer_fwd(arena, X, bf16=True, stream_bf16=True)
torch.cuda.synchronize()
lib = _lib.load()
buf = np.zeros(256 * 8, dtype=np.uint64)
rc = lib.pips_debug_token_trace(buf.ctypes.data_as(ctypes.c_void_p))
t = buf.reshape(256, 8).astype(np.int64)
d = np.diff(t[:, :6], axis=1)
names = ["tile requested, weights / biases loaded", "LayerNorm-1 statistics (incl. the tile's round trip)", "16 channel slots: LN-1 apply, 3 MFMAs, GELU, residual, stores",
         "LayerNorm-2 statistics", "LN-2 apply + stores issued"]
print("rc", rc, "-- last token-mix launch of a bf16 mixer pass (M = 16384, bf16 stream), wave 0 of 256 blocks; shader clocks (2.39 GHz), median over blocks")
for k, nm in enumerate(names):
    print("  %-66s %7.0f clk = %5.2f us   (min %6.0f, max %6.0f)" % (nm, np.median(d[:, k]), np.median(d[:, k]) / 2390.0, d[:, k].min(), d[:, k].max()))
print("  %-66s %7.0f clk = %5.2f us" % ("start -> stores issued", np.median(t[:, 5] - t[:, 0]), np.median(t[:, 5] - t[:, 0]) / 2390.0))
print("  first block start -> last block end %.2f us;  block starts spread over %.2f us" % ((t[:, 5].max() - t[:, 0].min()) / 2390.0, (t[:, 0].max() - t[:, 0].min()) / 2390.0))
