"""What stock MIOpen (torch.nn.functional.conv2d, fp32) reaches on the encoder's main conv shapes: a yardstick, not product code."""
import torch, torch.nn.functional as F
dev = "cuda:0"
torch.backends.cudnn.benchmark = True
def t(fn, n=20):
    for _ in range(5): fn()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (cin, cout, h, w, k, s) in ((64, 64, 184, 248, 3, 1), (64, 96, 184, 248, 3, 2), (96, 96, 92, 124, 3, 1), (128, 128, 46, 62, 3, 1),
                                (416, 256, 46, 62, 3, 1), (3, 64, 368, 496, 7, 2)):
    for cl in (False, True):
        x = torch.randn(8, cin, h, w, device=dev); wt = torch.randn(cout, cin, k, k, device=dev)
        if cl:
            x = x.contiguous(memory_format=torch.channels_last); wt = wt.contiguous(memory_format=torch.channels_last)
        us = t(lambda: F.conv2d(x, wt, None, stride=s, padding=k // 2))
        ho, wo = (h + 2 * (k // 2) - k) // s + 1, (w + 2 * (k // 2) - k) // s + 1
        fl = 2.0 * 8 * ho * wo * cout * cin * k * k
        print(f"{cin:3d}->{cout:3d} {h}x{w} k{k} s{s} {'NHWC' if cl else 'NCHW'}: {us:8.1f} us  {fl/us/1e6:6.1f} TFLOP/s")
