#!/usr/bin/env python
"""Tail-quantisation probe of k_conv_patch<128,8,3,0>: one 196->128 3x3 layer on 64x208 maps, batch swept so that the grid
goes from under one residency round (256 CUs x 4 workgroups) to a little over two.  Prints WGs, ms, TFLOP/s per batch."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cc_amd import ops


def timeit(fn, iters=8):
    fn(); fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    Cin, Cout, H, W = 196, 128, 64, 208
    w = torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.05
    for B in (4, 6, 8, 9, 10, 11, 12, 13, 14, 16, 18, 19, 20, 22, 24, 29, 30):
        x = torch.randn(B, Cin, H, W, device="cuda")
        t = timeit(lambda: ops._Conv2dFn.apply(x, w, None, None, 1, 1, 1, 1.0, 0.0))
        fl = 2.0 * B * H * W * Cin * Cout * 9
        tiles = B * (H // 8) * (W // 16)
        print("B=%2d  WGs=%5d  rounds=%.3f  %.3f ms  %.1f TF" % (B, tiles, tiles / 1024.0, t, fl / t / 1e9), flush=True)


if __name__ == "__main__":
    main()
