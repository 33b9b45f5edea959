#!/usr/bin/env python
"""What is the floor of a small convolution launch made of?  (Round 5: with the main loop of k_conv_patch ablated away completely the
direct-conv family still costs more than half of its time -- tools/ablate_conv.sh, profiles/r05_ab_round5.txt.)

Tiny-map layers of the CC step, 20 calls per captured hipGraph, time per call (main kernel + its split-K epilogue launch) while the
length of a workgroup's reduction chain is varied through the TOOLS build's planner switches (CC_CONV_MAXSPLIT: fewer splits = longer
chains; CC_CONV_TPS1: one tap per pipeline stage = three times the stages for the same work), beside a trivial launch of the same
library (2x up-sampling of a 4 x 1 x 2 x 7 map) as the floor of any launch.

    CC_LIB_PATH=tools/_bin/libccengine_tools.so python tools/floor_probe.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from tools import ab_env  # noqa: E402

ab_env.apply()
from cc_amd import ops, _lib  # noqa: E402

N = 20


def graph_time(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(N):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps / N * 1e3          # us per call


def main():
    E = _lib.engine()
    assert E.fn["cc_is_tools_build"]() == 1, "run with CC_LIB_PATH=tools/_bin/libccengine_tools.so"
    dev = torch.device("cuda")
    x0 = torch.randn(4, 1, 2, 7, device=dev)
    y0 = torch.empty(4, 1, 4, 14, device=dev)
    t = graph_time(lambda: E.call("cc_upsample2x_fwd", x0, y0, 4, 1, 2, 7, 14, 56, 1.0, _lib.STREAM))
    print("trivial launch (cc_upsample2x_fwd, 56 values): %.1f us per launch" % t)
    shapes = [(256, 256, 2, 7, 3), (512, 512, 2, 7, 3), (512, 512, 4, 13, 3), (256, 256, 8, 26, 3), (128, 256, 16, 52, 1), (512, 1024, 8, 26, 1)]
    for (Cin, Cout, H, W, k) in shapes:
        x = torch.randn(4, Cin, H, W, device=dev)
        w = torch.randn(Cout, Cin, k, k, device=dev) * 0.03
        b = torch.zeros(Cout, device=dev)
        row = []
        for tps1 in ("0", "1"):
            for ms in ("32", "8", "2", "1"):
                os.environ["CC_CONV_TPS1"] = tps1
                os.environ["CC_CONV_MAXSPLIT"] = ms
                os.environ["CC_NO_WINO"] = "1"              # the direct kernel on every shape
                with torch.no_grad():
                    t = graph_time(lambda: ops.conv2d(x, w, b, 1, k // 2, "relu"))
                row.append("%s%s %5.1f" % ("tps1 " if tps1 == "1" else "", "k<=" + ms, t))
        print("B4 C%d->M%d %dx%d %dx%d: %s   (us per call)" % (Cin, Cout, H, W, k, k, " | ".join(row)), flush=True)
        os.environ["CC_CONV_TPS1"] = "0"
        os.environ["CC_CONV_MAXSPLIT"] = "32"
        row = []
        for mb in ("0", "16", "64", "256", "1024"):          # channel tile halved while the launch has fewer blocks than this
            os.environ["CC_CONV_BM_MINBLOCKS"] = mb
            with torch.no_grad():
                t = graph_time(lambda: ops.conv2d(x, w, b, 1, k // 2, "relu"))
            row.append("minblocks %s %5.1f" % (mb, t))
        os.environ.pop("CC_CONV_BM_MINBLOCKS")
        print("        channel tile: %s" % " | ".join(row), flush=True)
    for k in ("CC_CONV_TPS1", "CC_CONV_MAXSPLIT", "CC_NO_WINO"):
        os.environ.pop(k, None)


if __name__ == "__main__":
    main()
