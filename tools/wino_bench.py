#!/usr/bin/env python
"""Winograd F(2x2, 3x3) kernel (wino.hip) against the direct implicit-GEMM kernel (conv.hip) on the 3x3 / stride-1 layer shapes
of the CC step: per shape the main device kernel's duration (HIP events inside the tools build, cc_timing_*), for the forward
convolution and the data-gradient, single problems and the grouped launches of Back2Future's parallel decoders, with the
per-step weight images (cc_repack_table) as the trainer uses them.  Also checks the two results against each other.

    python tools/wino_bench.py [--iters 5] [--quick]        (needs tools/_bin/libccengine_tools.so: python -m cc_amd.build --tools)
"""
import argparse
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from cc_amd import _lib, build, ops  # noqa: E402

PEAK = 157.3
SHAPES = [  # G, B, Cin, H, W, Cout
    (3, 4, 196, 64, 208, 128), (2, 4, 128, 64, 208, 196), (3, 4, 128, 64, 208, 128), (3, 4, 128, 64, 208, 96),
    (2, 4, 96, 64, 208, 128), (3, 4, 96, 64, 208, 64), (2, 4, 64, 64, 208, 96), (1, 4, 64, 64, 208, 64),
    (1, 4, 129, 64, 208, 64), (1, 4, 64, 64, 208, 129), (3, 4, 64, 64, 208, 32),
    (3, 4, 228, 32, 104, 128), (2, 4, 128, 32, 104, 228), (3, 4, 128, 32, 104, 128), (1, 4, 128, 32, 104, 128),
    (1, 4, 256, 32, 104, 128), (1, 4, 128, 32, 104, 256), (3, 4, 64, 32, 104, 64), (3, 4, 96, 32, 104, 64),
    (1, 4, 256, 16, 52, 256), (1, 4, 512, 16, 52, 256), (1, 4, 256, 16, 52, 512), (3, 4, 260, 16, 52, 128),
    (2, 4, 128, 16, 52, 260), (3, 4, 96, 16, 52, 96),
    (1, 4, 512, 8, 26, 512), (1, 4, 1024, 8, 26, 512), (1, 4, 512, 8, 26, 1024), (3, 4, 128, 8, 26, 128),
    (1, 4, 512, 4, 13, 512), (3, 4, 192, 4, 13, 192),
    (1, 4, 32, 128, 416, 32), (1, 4, 65, 128, 416, 32), (1, 4, 32, 128, 416, 65), (1, 4, 64, 128, 416, 32),
]
QUICK = [SHAPES[0], SHAPES[2], SHAPES[7], SHAPES[14], SHAPES[19], SHAPES[25]]


def collect(eng):
    buf = ctypes.create_string_buffer(1 << 18)
    n = eng.fn["cc_timing_collect"](ctypes.addressof(buf), 1 << 18)
    out = {}
    for ln in buf.raw[:n].decode().splitlines():
        nm, cnt, ms, gf = ln.split("\t")
        out[nm] = (int(cnt), float(ms), float(gf))
    return out


def run(eng, G, B, Cin, H, W, Cout, iters, wino):
    """-> (forward ms, dgrad ms per call of the MAIN kernel [+ split-K epilogue listed separately], outputs)"""
    os.environ["CC_NO_WINO"] = "0" if wino else "1"
    ops.packs.reset()
    g = torch.Generator(device="cuda").manual_seed(1)
    xs = [torch.randn(B, Cin, H, W, device="cuda", generator=g).requires_grad_(True) for _ in range(G)]
    ws = [(torch.randn(Cout, Cin, 3, 3, device="cuda", generator=g) * 0.05).requires_grad_(True) for _ in range(G)]
    bs = [torch.randn(Cout, device="cuda", generator=g) for _ in range(G)]

    def step():
        if G == 1:
            ys = [ops.conv2d(xs[0], ws[0], bs[0], 1, 1, "lrelu")]
        else:
            ys = ops.conv2d_group(xs, ws, bs, 1, 1, "lrelu")
        gx = torch.autograd.grad(ys, xs, [torch.ones_like(y) for y in ys])
        return ys, gx
    ops.packs.recording = True
    step()                                   # registers the layers
    ops.packs.prepack_all()
    out = step()                             # first launches of this code path stay untimed
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        step()
    e1.record()
    torch.cuda.synchronize()
    call_ms = e0.elapsed_time(e1) / iters          # forward + data-gradient calls incl. split-K epilogues and host gaps
    eng.call("cc_timing_enable", 1)
    for _ in range(iters):
        step()
    torch.cuda.synchronize()
    recs = collect(eng)
    ops.packs.invalidate()
    return recs, out, call_ms


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--quick", action="store_true")
    args = ap.parse_args()
    lib = build.TOOLS_OUT if os.path.isfile(build.TOOLS_OUT) else build.build_tools()
    os.environ["CC_TIMING_DETAIL"] = "1"
    with _lib.use_library(lib) as eng:
        assert eng.fn["cc_is_tools_build"]() == 1
        print("%-34s %9s %9s %7s | %9s %9s %7s | %s" % ("shape (G B Cin HxW Cout)", "direct ms", "wino ms", "x", "dir TF", "wino TFeq",
                                                        "MFMA TF", "max rel diff fwd / frac(dgrad off > 1e-4)"))
        tot_d = tot_w = 0.0
        for (G, B, Cin, H, W, Cout) in (QUICK if args.quick else SHAPES):
            rd, od, cd = run(eng, G, B, Cin, H, W, Cout, args.iters, False)
            rw, ow, cw = run(eng, G, B, Cin, H, W, Cout, args.iters, True)
            # conv kernels only (the weight-gradient records do not occur: only the inputs require gradients' data-gradient)
            md = sum(ms for nm, (c, ms, gf) in rd.items() if nm.startswith("k_conv") or nm.startswith("k_wino")) / args.iters
            mw = sum(ms for nm, (c, ms, gf) in rw.items() if nm.startswith("k_conv") or nm.startswith("k_wino")) / args.iters
            used = any(nm.startswith("k_wino") for nm in rw)
            gf_direct = 2 * 2e-9 * G * B * H * W * Cout * Cin * 9            # forward + data-gradient
            gf_mfma = sum(gf for nm, (c, ms, gf) in rw.items() if nm.startswith("k_wino")) / args.iters
            e1 = max(float((a.detach() - b.detach()).abs().max() / b.detach().abs().max()) for a, b in zip(ow[0], od[0]))
            # data-gradient: the FRACTION of elements off by more than 1e-4 of the maximum (a LeakyReLU derivative flips where the two
            # forward results straddle zero: a handful of elements among millions, each O(1))
            e2 = max(float(((a - b).abs() > 1e-4 * b.abs().max()).float().mean()) for a, b in zip(ow[1], od[1]))
            tot_d += md
            tot_w += mw
            print("%d %d %4d %3dx%-3d %4d %12s %9.3f %9.3f %7.2f | %9.1f %9.1f %7.1f | %.1e %.1e | calls %.3f -> %.3f ms" %
                  (G, B, Cin, H, W, Cout, "" if used else "(direct)", md, mw, md / mw, gf_direct / md, gf_direct / mw,
                   gf_mfma / mw if used else 0.0, e1, e2, cd, cw), flush=True)
            names = sorted(set(nm.split(" ")[0] for nm in rw if nm.startswith("k_")))
            print("      kernels: %s" % ", ".join("%s %.3f" % (nm.split(" ")[0] + ("+" + nm.split(" k")[-1].split("]")[0] if " k" in nm else ""),
                                                              ms / args.iters) for nm, (c, ms, gf) in rw.items()), flush=True)
        print("total: direct %.3f ms, winograd %.3f ms" % (tot_d, tot_w))


if __name__ == "__main__":
    main()
