#!/usr/bin/env python
"""Which Winograd instance for which layer shape?  Forward + data-gradient of the 3x3 / stride-1 layers of the <= 64x208 levels,
captured in a hipGraph (as the trainer runs them) and replayed, per planner variant of the TOOLS build:
    r4      CC_WINO_SMALL=0          64 x 64 blocks, split-K + epilogue launch where the planner of round 4 says so
    s1      32 x 32, four waves (two workgroups per CU), whole reduction or sliced by its cost model
    s1full  ... the whole reduction forced (no slices: no epilogue launch)
    s2      32 x 32, eight waves, the reduction halved inside the workgroup
    auto    the shipped cost model
Time = graph replay of (forward, data-gradient) / number of replays: includes every epilogue / padding launch of the calls.

    python tools/wino_tile_probe.py [--iters 20]         (needs tools/_bin/libccengine_tools.so)
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from cc_amd import _lib, build, ops  # noqa: E402

SHAPES = [  # G, B, Cin, H, W, Cout
    (1, 4, 64, 64, 208, 64), (1, 4, 129, 64, 208, 64), (2, 4, 96, 64, 208, 64), (3, 4, 64, 64, 208, 32), (2, 4, 128, 64, 208, 128),
    (1, 4, 128, 32, 104, 128), (1, 4, 256, 32, 104, 128), (1, 4, 128, 32, 104, 256), (3, 4, 128, 32, 104, 128), (3, 4, 64, 32, 104, 64),
    (2, 4, 128, 32, 104, 96), (3, 4, 228, 32, 104, 128),
    (1, 4, 256, 16, 52, 256), (1, 4, 512, 16, 52, 256), (1, 4, 256, 16, 52, 512), (3, 4, 260, 16, 52, 128), (3, 4, 96, 16, 52, 96),
    (3, 4, 128, 16, 52, 128), (2, 4, 128, 16, 52, 128),
    (1, 4, 512, 8, 26, 512), (1, 4, 1024, 8, 26, 512), (3, 4, 128, 8, 26, 128), (1, 4, 512, 4, 13, 512),
    (1, 4, 32, 128, 416, 32), (1, 4, 64, 128, 416, 32),
]
VARIANTS = [
    ("r4", {"CC_WINO_SMALL": "0"}),
    ("s1", {"CC_WINO_SMALL": "2", "CC_WINO_S_TILE": "1"}),
    ("s1full", {"CC_WINO_SMALL": "2", "CC_WINO_S_TILE": "1", "CC_WINO_S_STAGE": "1", "CC_WINO_S_ALONE": "1"}),
    ("s2", {"CC_WINO_SMALL": "2", "CC_WINO_S_TILE": "2"}),
    ("auto", {}),
]
KEYS = sorted({k for _, e in VARIANTS for k in e} | {"CC_WINO_MINM", "CC_WINO_MINC"})


def run(G, B, Cin, H, W, Cout, iters, env):
    for k in KEYS:
        os.environ.pop(k, None)
    os.environ.update(env)
    ops.packs.reset()
    g = torch.Generator(device="cuda").manual_seed(1)
    xs = [torch.randn(B, Cin, H, W, device="cuda", generator=g).requires_grad_(True) for _ in range(G)]
    ws = [(torch.randn(Cout, Cin, 3, 3, device="cuda", generator=g) * 0.05).requires_grad_(True) for _ in range(G)]
    bs = [torch.randn(Cout, device="cuda", generator=g) for _ in range(G)]
    gy = [torch.randn(B, Cout, H, W, device="cuda", generator=g) for _ in range(G)]

    def step():
        if G == 1:
            ys = [ops.conv2d(xs[0], ws[0], bs[0], 1, 1, "lrelu")]
        else:
            ys = ops.conv2d_group(xs, ws, bs, 1, 1, "lrelu")
        gx = torch.autograd.grad(ys, xs, gy)
        return ys, gx
    ops.packs.recording = True
    step()
    ops.packs.prepack_all()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        out = step()
        step()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(4):
            step()
    gr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        gr.replay()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / (4 * iters)
    res = [t.detach().clone() for t in out[0]] + [t.detach().clone() for t in out[1]]
    ops.packs.invalidate()
    del gr
    return ms, res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    args = ap.parse_args()
    lib = build.TOOLS_OUT if os.path.isfile(build.TOOLS_OUT) else build.build_tools()
    with _lib.use_library(lib) as eng:
        assert eng.fn["cc_is_tools_build"]() == 1
        print("%-30s %s   GFLOP(exec) | best  (us per forward + data-gradient pair; max |diff| vs r4 relative to max |r4|)" %
              ("shape (G B Cin HxW Cout)", " ".join("%8s" % n for n, _ in VARIANTS)))
        tot = {n: 0.0 for n, _ in VARIANTS}
        for (G, B, Cin, H, W, Cout) in SHAPES:
            row, ref, worst = [], None, 0.0
            for name, env in VARIANTS:
                ms, res = run(G, B, Cin, H, W, Cout, args.iters, env)
                if ref is None:
                    ref = res
                else:
                    for a, b in zip(res[:G], ref[:G]):           # forward outputs (the data-gradients differ where LeakyReLU' flips)
                        worst = max(worst, float((a - b).abs().max() / b.abs().max()))
                row.append(ms)
                tot[name] += ms
            gf = 2 * 2e-9 * 16 * G * B * ((H + 1) // 2) * ((W + 1) // 2) * Cout * Cin
            best = min(range(len(row)), key=lambda i: row[i])
            print("%d %d %4d %3dx%-3d %4d %9s %s %9.2f | %-6s %.1e" % (G, B, Cin, H, W, Cout, "", " ".join("%8.1f" % (1e3 * m) for m in row), gf,
                                                                   VARIANTS[best][0], worst), flush=True)
        print("total %s" % " ".join("%s %.3f ms" % (n, tot[n]) for n, _ in VARIANTS))


if __name__ == "__main__":
    main()
