"""The clock the two generated bf16 GEMMs of the mixer run at (variant build: sh tools/build_variant.sh t4clk gemm_bf16_t4 -DPIPS_T4_CLOCK;
PIPS_LIB_PATH=build/libpips_t4clk.so): every wave stamps s_memtime and s_memrealtime around the generated statement."""
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
for n in (3, 300):
    for _ in range(n):
        ops.mixer_fwd(arena, X, bf16=True, stream_bf16=True)
    torch.cuda.synchronize()
    buf = np.zeros(2 * 1024 * 2, dtype=np.uint64)
    rc = lib.pips_debug_t4_clock(buf.ctypes.data_as(ctypes.c_void_p))
    t = buf.reshape(2, 1024, 2).astype(np.float64)
    for k, name in enumerate(("gemm_bf16_t4_gelu_kernel (up-projection, 34.4 GFLOP, 2 097 152 MFMAs)", "gemm_bf16_t4_res_kernel<true> (down-projection, 34.4 GFLOP)")):
        ghz = t[k, :, 0] / np.maximum(t[k, :, 1], 1) / 10.0
        us = t[k, :, 1] / 100.0
        print("after %3d passes  %-72s shader clock %.2f GHz (min %.2f, max %.2f);  a wave: %6.0f clocks = %.2f us of real time (min %.2f, max %.2f);  its 2 048 MFMAs alone: %.1f us at that clock"
              % (n, name, np.median(ghz), ghz.min(), ghz.max(), np.median(t[k, :, 0]), np.median(us), us.min(), us.max(), 2048 * 16.4 / np.median(ghz) / 1e3))
