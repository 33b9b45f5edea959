#!/usr/bin/env python
"""Round 4: why do the data-gradients of the prediction heads (3x3, 1 / 4 output channels -> reduction length 1 / 4) on the small
maps take 30-40 us each?  Times cc_conv2d_dgrad alone (20 calls per hipGraph replay) over the reduction length."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cc_amd import ops
from cc_amd._lib import engine

CASES = [  # Cin (= channels of the gradient produced), H, W
    (512, 8, 26), (256, 16, 52), (128, 32, 104), (64, 64, 208),
]
COUTS = [1, 4, 8, 16, 64]


def main():
    dev = torch.device("cuda")
    E = engine()
    B, R, S, stride, pad = 4, 3, 3, 1, 1
    n = 20
    for Cin, H, W in CASES:
        for Cout in COUTS:
            gy = torch.randn(B, Cout, H, W, device=dev)
            w = torch.randn(Cout, Cin, R, S, device=dev) * 0.05
            gx = torch.empty(B, Cin, H, W, device=dev)
            ws = ops._ws(E.call("cc_conv2d_dgrad_ws_bytes", B, Cout, H, W, Cin, R, S, stride, pad, H, W), gy)
            pk = ops.packs.get("dgrad", w, (B, Cout, H, W, Cin, R, S, stride, pad, H, W, Cin * R * S, R * S))

            def call():
                E.call("cc_conv2d_dgrad", gy, w, None, gx, ws, pk, B, Cout, H, W, Cout * H * W, Cin, R, S, stride, pad, H, W,
                       Cin * H * W, Cin * R * S, R * S, 0, 1.0, 0.0, ops.STREAM)
            call()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for _ in range(n):
                    call()
            g.replay()
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(5):
                g.replay()
            e.record()
            torch.cuda.synchronize()
            us = s.elapsed_time(e) * 1e3 / (5 * n)
            print("dgrad B%d dX[%d,%dx%d] <- dY %d ch: %.1f us/call" % (B, Cin, H, W, Cout, us), flush=True)


if __name__ == "__main__":
    main()
