#!/usr/bin/env python
"""Is there HBM channel camping when 16-32 NCHW channel planes of 2^16 * 13 bytes are walked in lock step?
Times the thin wgrad (16 planes of dY + 16 of X read at the same pixel offsets) and a conv forward on maps whose plane size
is / is not a multiple of large powers of two."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cc_amd._lib import engine, STREAM
E = engine()

def wgrad(B, M, Cin, H, W):
    a = torch.randn(B, M, H, W, device="cuda"); x = torch.randn(B, Cin, H, W, device="cuda")
    gw = torch.zeros(M, Cin, 3, 3, device="cuda")
    ws = torch.empty(E.call("cc_conv2d_wgrad_ws_bytes", B, M, H, W, Cin, 3, 3, 1) // 4 + 64, device="cuda")
    def run():
        E.call("cc_conv2d_wgrad", a, x, gw, ws, B, M, H, W, M * H * W, Cin, H, W, Cin * H * W, 3, 3, 1, 1, Cin * 9, 9, 0, STREAM)
    return run

def fwd(B, Cin, Cout, H, W):
    x = torch.randn(B, Cin, H, W, device="cuda"); w = torch.randn(Cout, Cin, 3, 3, device="cuda")
    y = torch.empty(B, Cout, H, W, device="cuda")
    ws = torch.empty(E.call("cc_conv2d_fwd_ws_bytes", B, Cin, H, W, Cout, 3, 3, 1, 1, H, W) // 4 + 64, device="cuda")
    def run():
        E.call("cc_conv2d_fwd", x, w, None, None, y, ws, None, B, Cin, H, W, Cin * H * W, Cout, 3, 3, 1, 1, H, W, Cout * H * W, 0, 0, 1.0, 0.0, STREAM)
    return run

def timeit(run, n=10):
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

for (H, W) in ((256, 832), (256, 848), (256, 816), (264, 832), (248, 832), (256, 800), (128, 416), (128, 432), (64, 208), (64, 224)):
    t = timeit(wgrad(4, 16, 16, H, W))
    px = 4 * H * W
    t2 = timeit(fwd(4, 128, 128, H // 4, W // 4)) if H >= 256 else 0
    print("%4dx%-4d plane %8d B (= 2^%d * %d)  thin wgrad 16x16: %.4f ms  %.2f ns/kpx   | fwd 128->128 @%dx%d %.4f ms %.1f TF" % (
        H, W, H * W * 4, ((H * W * 4) & -(H * W * 4)).bit_length() - 1, (H * W * 4) // ((H * W * 4) & -(H * W * 4)), t, 1e6 * t / (px / 1e3),
        H // 4, W // 4, t2, (2.0 * 4 * (H // 4) * (W // 4) * 128 * 128 * 9 / t2 / 1e9) if t2 else 0), flush=True)
