#!/usr/bin/env python
"""Winograd F(3x3, 2x2) weight-gradient kernel (wino_wgrad.hip) against the kernels it replaces (k_wgrad3x3 / k_wgrad_thin / k_wgrad)
on the 3x3 / stride-1 layer shapes of the CC step: main-kernel durations from the tools build's timing registry + the call time
(kernel + reduction), results compared.   python tools/wino_wgrad_bench.py [--iters 5] [--quick]"""
import argparse
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from cc_amd import _lib, build, ops  # noqa: E402
from wino_bench import SHAPES, QUICK, collect  # noqa: E402


def run(eng, G, B, Cin, H, W, Cout, iters, wino):
    os.environ["CC_NO_WINO_WGRAD"] = "0" if wino else "1"
    g = torch.Generator(device="cuda").manual_seed(1)
    E = eng
    a = [torch.randn(B, Cout, H, W, device="cuda", generator=g) for _ in range(G)]
    x = [torch.randn(B, Cin, H, W, device="cuda", generator=g) for _ in range(G)]
    gw = [torch.zeros(Cout, Cin, 3, 3, device="cuda") for _ in range(G)]

    def step():
        for t in gw:
            t.zero_()
        ops._wgrad_group(a, x, gw, x[0], B, Cout, H, W, Cout * H * W, Cin, H, W, Cin * H * W, 3, 3, 1, 1, Cin * 9, 9, 0)
    step()
    step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        step()
    e1.record()
    torch.cuda.synchronize()
    call_ms = e0.elapsed_time(e1) / iters
    E.call("cc_timing_enable", 1)
    for _ in range(iters):
        step()
    torch.cuda.synchronize()
    recs = collect(E)
    return recs, [t.clone() for t in gw], call_ms


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--quick", action="store_true")
    args = ap.parse_args()
    os.environ["CC_TIMING_DETAIL"] = "1"
    with _lib.use_library(build.TOOLS_OUT) as eng:
        print("%-30s %10s %10s %6s | %8s %9s %8s | %s" % ("shape (G B Cin HxW Cout)", "before ms", "wino ms", "x", "before TF", "wino TFeq",
                                                       "MFMA TF", "rel diff | calls (kernel + reduce) ms | kernels"))
        td = tw = 0.0
        for (G, B, Cin, H, W, Cout) in (QUICK if args.quick else SHAPES):
            rd, gd, cd = run(eng, G, B, Cin, H, W, Cout, args.iters, False)
            rw, gwv, cw = run(eng, G, B, Cin, H, W, Cout, args.iters, True)
            md = sum(ms for nm, (c, ms, gf) in rd.items()) / args.iters
            mw = sum(ms for nm, (c, ms, gf) in rw.items()) / args.iters
            used = any(nm.startswith("k_wino_wgrad") for nm in rw)
            gf_direct = 2e-9 * G * B * H * W * Cout * Cin * 9
            gf_mfma = sum(gf for nm, (c, ms, gf) in rw.items() if nm.startswith("k_wino")) / args.iters
            err = max(float((p - q).abs().max() / q.abs().max()) for p, q in zip(gwv, gd))
            td += cd
            tw += cw
            print("%d %d %4d %3dx%-3d %4d %8s %10.3f %10.3f %6.2f | %8.1f %9.1f %8.1f | %.1e | %.3f -> %.3f | %s -> %s" %
                  (G, B, Cin, H, W, Cout, "" if used else "(same)", md, mw, md / mw, gf_direct / md, gf_direct / mw,
                   gf_mfma / mw if used else 0.0, err, cd, cw, ",".join(sorted(set(n.split(" ")[0] for n in rd))),
                   ",".join(sorted(set(n.split(" ")[0] for n in rw)))), flush=True)
        print("total call time: before %.3f ms, winograd %.3f ms" % (td, tw))


if __name__ == "__main__":
    main()
