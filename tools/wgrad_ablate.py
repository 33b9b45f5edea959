#!/usr/bin/env python
"""Ablation of the per-tap wgrad kernel on the B2F 128->128 3x3 @64x208 layer (B=4): full / no-DMA / no-MFMA, and the
im2col-style kernel, timed with HIP events."""
import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    import torch
    from cc_amd._lib import engine, STREAM
    E = engine()
    shapes = {"b2f128": (4, 128, 64, 208, 128, 3, 1, 1), "iconv2": (4, 32, 128, 416, 65, 3, 1, 1), "conv5": (4, 512, 8, 26, 512, 3, 1, 1),
              "c1_2": (4, 32, 128, 416, 32, 7, 1, 3), "conv2": (4, 64, 64, 208, 64, 3, 1, 1), "conv3": (4, 128, 32, 104, 128, 3, 1, 1),
              "dec0": (4, 128, 64, 208, 196, 3, 1, 1), "ic1": (4, 16, 256, 832, 17, 3, 1, 1), "dec6": (4, 64, 64, 208, 96, 3, 1, 1)}
    B, M, AH, AW, Cin, R, si, pad = shapes[sys.argv[1]]
    IH, IW = AH * si, AW * si
    a = torch.randn(B, M, AH, AW, device="cuda"); x = torch.randn(B, Cin, IH, IW, device="cuda")
    gw = torch.empty(M, Cin, R, R, device="cuda")
    ws = torch.empty(E.call("cc_conv2d_wgrad_ws_bytes", B, M, AH, AW, Cin, R, R, si) // 4 + 64, device="cuda")
    def run():
        E.call("cc_conv2d_wgrad", a, x, gw, ws, B, M, AH, AW, M * AH * AW, Cin, IH, IW, Cin * IH * IW, R, R, si, pad, Cin * R * R, R * R, 0, STREAM)
    run(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): run()
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 10
    fl = 2.0 * B * AH * AW * M * Cin * R * R
    import torch.nn.functional as F
    err = -1.0
    if R == 3 and si == 1:
        xr = x.clone().requires_grad_(True); wr = torch.zeros(M, Cin, R, R, device="cuda", requires_grad=True)
        F.conv2d(xr, wr, None, 1, pad).backward(a)
        err = float((gw - wr.grad).abs().max() / wr.grad.abs().max())
    print("%-8s nbuf=%s split=%s dbg=%s  %.3f ms  %.1f TF  relerr %.2e" % (sys.argv[1], os.environ.get("CC_W3_NBUF", "-"), os.environ.get("CC_W3_SPLIT", "-"), os.environ.get("CC_W3_DBG", "0"), ms, fl / ms / 1e9, err))
else:
    for shape in ("b2f128", "dec0", "dec6", "conv2", "conv3", "iconv2"):
        for env in ({"CC_NO_WGRAD3X3": "1"}, {"CC_W3_DBG": "0"}, {"CC_W3_DBG": "1"}, {"CC_W3_SPLIT": "1024"}):
            subprocess.run([sys.executable, __file__, shape], env=dict(os.environ, **env))
