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
lib = _lib.load()
PASSES = int(os.environ.get("PIPS_TRACE_PASSES", "3"))


def trace(passes):
    for _ in range(passes):
        ops.mixer_fwd(arena, X, bf16=True, stream_bf16=True)
    torch.cuda.synchronize()
    buf = np.zeros(256 * 8, dtype=np.uint64)
    rc = lib.pips_debug_token_trace(buf.ctypes.data_as(ctypes.c_void_p))
    return rc, buf.reshape(256, 8).astype(np.int64)


# the clock after 3, 30 and 300 back-to-back mixer passes (1.4 ms each): does it depend on how long the device has been busy?
for n in (3, 30, 300):
    _, tt = trace(n)
    g = (tt[:, 5] - tt[:, 0]) / np.maximum(tt[:, 7] - tt[:, 6], 1) / 10.0
    print("after %3d more passes: shader clock %.2f GHz (min %.2f, max %.2f), wave %.2f us real time" %
          (n, np.median(g), g.min(), g.max(), np.median(tt[:, 7] - tt[:, 6]) / 100.0))
rc, t = trace(PASSES)
d = np.diff(t[:, :6], axis=1)
names = ["tile requested, weights / biases loaded", "LayerNorm-1 statistics (incl. the tile's round trip)", "16 channel slots: LN-1 apply, 3 MFMAs, GELU, residual, stores",
         "LayerNorm-2 statistics", "LN-2 apply + stores issued"]
rt = t[:, 6:8]                                        # s_memrealtime at stamps 0 and 5: 100 MHz, the same counter on every compute unit
ghz = (t[:, 5] - t[:, 0]) / np.maximum(rt[:, 1] - rt[:, 0], 1) / 10.0
mhz = float(np.median(ghz)) * 1000.0
print("rc", rc, "-- last token-mix launch of a bf16 mixer pass (M = 16384, bf16 stream), wave 0 of 256 blocks; shader clocks (us at the measured %.2f GHz), median over blocks" % (mhz / 1000.0))
for k, nm in enumerate(names):
    print("  %-66s %7.0f clk = %5.2f us   (min %6.0f, max %6.0f)" % (nm, np.median(d[:, k]), np.median(d[:, k]) / mhz, d[:, k].min(), d[:, k].max()))
print("  %-66s %7.0f clk = %5.2f us" % ("start -> stores issued", np.median(t[:, 5] - t[:, 0]), np.median(t[:, 5] - t[:, 0]) / mhz))
print("  shader clock while the kernel ran (s_memtime / s_memrealtime per wave): median %.2f GHz (min %.2f, max %.2f)" % (np.median(ghz), ghz.min(), ghz.max()))
print("  real time: wave start -> stores issued median %.2f us;  first start -> last stores issued %.2f us;  starts spread over %.2f us" %
      (np.median(rt[:, 1] - rt[:, 0]) / 100.0, (rt[:, 1].max() - rt[:, 0].min()) / 100.0, (rt[:, 0].max() - rt[:, 0].min()) / 100.0))
