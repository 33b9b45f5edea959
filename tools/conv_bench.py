#!/usr/bin/env python
"""Per-layer timing of the fp32 MFMA conv kernels on representative CC layer shapes,
beside torch/MIOpen on the same tensors.  Prints one line per (layer, pass): ms, TFLOP/s, fraction of 157.3."""
import json
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from cc_amd import ops, config

PEAK = 157.3
LAYERS = [  # name, B, Cin, H, W, Cout, k, stride, pad, transposed
    ("disp.conv1.0   3->32 k7 s2", 4, 3, 256, 832, 32, 7, 2, 3, False),
    ("disp.conv1.2  32->32 k7   ", 4, 32, 128, 416, 32, 7, 1, 3, False),
    ("disp.conv2    64->64 k3   ", 4, 64, 64, 208, 64, 3, 1, 1, False),
    ("disp.conv3   128->128 k3  ", 4, 128, 32, 104, 128, 3, 1, 1, False),
    ("disp.conv5   512->512 k3  ", 4, 512, 8, 26, 512, 3, 1, 1, False),
    ("disp.iconv6 1024->512 k3  ", 4, 1024, 8, 26, 512, 3, 1, 1, False),
    ("disp.iconv2   65->32 k3   ", 4, 65, 128, 416, 32, 3, 1, 1, False),
    ("disp.iconv1   17->16 k3   ", 4, 17, 256, 832, 16, 3, 1, 1, False),
    ("disp.head     16->1  k3   ", 4, 16, 256, 832, 1, 3, 1, 1, False),
    ("disp.upconv1  32->16 T k3 ", 4, 32, 128, 416, 16, 3, 2, 1, True),
    ("b2f.dec2.0   196->128 k3  ", 4, 196, 64, 208, 128, 3, 1, 1, False),
    ("b2f.dec2.2   128->128 k3  ", 4, 128, 64, 208, 128, 3, 1, 1, False),
    ("b2f.dec2.6    96->64  k3  ", 4, 96, 64, 208, 64, 3, 1, 1, False),
    ("b2f.dec2.10   32->2   k3  ", 4, 32, 64, 208, 2, 3, 1, 1, False),
    ("b2f.feat1     3->16 k3 s2 ", 4, 3, 256, 832, 16, 3, 2, 1, False),
    ("mask.conv1   15->16 k7 s2 ", 4, 15, 256, 832, 16, 7, 2, 3, False),
    ("mask.deconv1  48->16 T k4 ", 4, 48, 128, 416, 16, 4, 2, 1, True),
    ("mask.head     16->4  k3   ", 4, 16, 256, 832, 4, 3, 1, 1, False),
]


def timeit(fn, iters=5):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    dev = "cuda"
    rows = []
    for (name, B, Cin, H, W, Cout, k, st, pad, tr) in LAYERS:
        x = torch.randn(B, Cin, H, W, device=dev)
        if tr:
            w = torch.randn(Cin, Cout, k, k, device=dev) * 0.05
            op = 1 if k == 3 else 0
            fwd_hip = lambda: ops._ConvT2dFn.apply(x, w, None, st, pad, op, 1)
            fwd_ref = lambda: F.relu(F.conv_transpose2d(x, w, None, st, pad, op))
            macs = B * H * W * Cin * Cout * k * k
        else:
            w = torch.randn(Cout, Cin, k, k, device=dev) * 0.05
            fwd_hip = lambda: ops._Conv2dFn.apply(x, w, None, None, st, pad, 1, 1.0, 0.0)
            fwd_ref = lambda: F.relu(F.conv2d(x, w, None, st, pad))
            OH, OW = (H + 2 * pad - k) // st + 1, (W + 2 * pad - k) // st + 1
            macs = B * OH * OW * Cin * Cout * k * k
        fl = 2.0 * macs
        err = float((fwd_hip() - fwd_ref()).abs().max() / fwd_ref().abs().max())
        t_h = timeit(fwd_hip)
        t_r = timeit(fwd_ref)
        xg = x.clone().requires_grad_(True)
        wg = w.clone().requires_grad_(True)

        def fb(f_hip):
            if tr:
                y = ops._ConvT2dFn.apply(xg, wg, None, st, pad, op, 1) if f_hip else F.relu(F.conv_transpose2d(xg, wg, None, st, pad, op))
            else:
                y = ops._Conv2dFn.apply(xg, wg, None, None, st, pad, 1, 1.0, 0.0) if f_hip else F.relu(F.conv2d(xg, wg, None, st, pad))
            gy = torch.ones_like(y)
            torch.autograd.grad(y, [xg, wg], gy)
        t_hb = timeit(lambda: fb(True), 3)
        t_rb = timeit(lambda: fb(False), 3)
        row = dict(layer=name.strip(), relerr=round(err, 8), gflop=round(fl / 1e9, 2), hip_fwd_ms=round(t_h, 3), miopen_fwd_ms=round(t_r, 3),
                   hip_fwd_tf=round(fl / t_h / 1e9, 1), miopen_fwd_tf=round(fl / t_r / 1e9, 1),
                   hip_fwdbwd_ms=round(t_hb, 3), miopen_fwdbwd_ms=round(t_rb, 3),
                   hip_fwdbwd_tf=round(3 * fl / t_hb / 1e9, 1), miopen_fwdbwd_tf=round(3 * fl / t_rb / 1e9, 1))
        rows.append(row)
        print(json.dumps(row), flush=True)
    tot_h = sum(r["hip_fwdbwd_ms"] for r in rows)
    tot_r = sum(r["miopen_fwdbwd_ms"] for r in rows)
    print("TOTAL fwd+bwd ms  hip %.2f  miopen %.2f" % (tot_h, tot_r))


if __name__ == "__main__":
    main()
