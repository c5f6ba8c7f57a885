"""Half-step timeline of conv3x3_c64_pp_kernel (a -DPIPS_PP_TRACE build: sh tools/build_variant.sh or the recipe below).
    hipcc ... -DPIPS_PP_TRACE -> tools/libpips_pptrace.so;  PIPS_LIB_PATH=tools/libpips_pptrace.so python tools/conv_pp_trace.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import _tunelib  # noqa: F401  (PIPS_LIB_PATH -> pips_amd._lib.use_library)
from pips_amd import ops, _lib
lib = _lib.load()
dev = "cuda:0"
F_, H, W = int(os.environ.get('PP_F', '64')), 184, 248
g = torch.Generator().manual_seed(0)
x = torch.randn(F_, H, W, 64, generator=g).bfloat16().to(dev)
w = (torch.randn(64, 3, 3, 64, generator=g) / 24).bfloat16().to(dev)
b = torch.randn(64, generator=g).to(dev)
nrm = torch.stack([torch.randn(F_, 64, generator=g) * 0.3, torch.rand(F_, 64, generator=g) + 0.5], -1).to(dev)
for norm in (None, nrm):
    for _ in range(3):
        ops.conv_nhwc_bf16_maps(x, w, b, 3, 1, 1, in_norm=norm, want_stats=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.conv_nhwc_bf16_maps(x, w, b, 3, 1, 1, in_norm=norm, want_stats=True)
    e1.record(); e1.synchronize()
    print("norm-on-load" if norm is not None else "plain", "%.1f us per launch" % (e0.elapsed_time(e1) * 100))
out_buf = torch.empty(F_, H, W, 64, dtype=torch.bfloat16, device=dev)
cap = ((W + 31) // 32) * ((H + 3) // 4) * 4
stats_buf = torch.zeros(F_, cap, 64, 4, device=dev)
tiles = ctypes.c_int(0)
def launch(norm):
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    rc = lib.pips_conv_nhwc_bf16_maps(_lib.ptr(x), _lib.ptr(norm), F_, H, W, 64, _lib.ptr(w), _lib.ptr(b), 64, 3, 1, 1, _lib.ptr(out_buf), 1,
                                      _lib.ptr(stats_buf), cap, ctypes.byref(tiles), st)
    assert rc == 0
def bench(norm, n=20):
    """the kernel alone: preallocated output and statistics buffers"""
    for _ in range(3):
        launch(norm)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        launch(norm)
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print("kernel alone: plain %.1f us, norm-on-load %.1f us" % (bench(None), bench(nrm)))
if hasattr(lib, "pips_pp_dbg"):
    lib.pips_pp_dbg.argtypes = [ctypes.c_int]
    for bits, what in ((0, "full"), (1, "no epilogue stores"), (2, "no staging writes"), (4, "no fetch loads"), (8, "no MFMA phase"),
                       (16, "no epilogue"), (1 | 4, "no loads, no stores"), (8 | 16, "no MFMA, no epilogue"), (2 | 4 | 8 | 16, "barriers + stage VALU only"),
                       (31, "empty"), (31 | 64, "empty, no stage VALU"), (31 | 64 | 128, "barriers + loop only")):
        lib.pips_pp_dbg(bits | 32)
        print(f"ablation {bits:2d} ({what}): {bench(nrm):.1f} us")
    lib.pips_pp_dbg(0)
if hasattr(lib, "pips_pp_trace"):
    lib.pips_pp_trace.argtypes = [ctypes.c_void_p]
    tr = torch.zeros(8, 2, 64, 4, dtype=torch.int64, device=dev)
    assert lib.pips_pp_trace(tr.data_ptr()) == 0
    ops.conv_nhwc_bf16_maps(x, w, b, 3, 1, 1, in_norm=nrm, want_stats=True)
    torch.cuda.synchronize()
    lib.pips_pp_trace(None)
    t = tr.cpu()
    for blk in (0, 5):
        print("block", blk)
        for grp in range(2):
            for h in range(4, 16):
                a, m, e, kind = [int(v) for v in t[blk, grp, h]]
                if kind == 0: continue
                nxt = int(t[blk, grp, h + 1, 0])
                print(f"  grp {grp} h {h:2d} {'L' if kind == 1 else 'C'}: {'epilogue' if kind == 1 else 'fetch   '} {m - a:6d}  {'stage  ' if kind == 1 else 'compute'} {e - m:6d}  wait@barrier {nxt - e:6d}")
