#!/usr/bin/env python
"""Do two independent chains of under-filled kernels overlap on this runtime -- eagerly on two HIP streams, and as two branches of one
captured hipGraph?  (Rounds 1 / 2 found that hipGraph branches did not overlap; re-checked in round 5 with the 32x32 Winograd
kernel: 208-416 workgroups per launch, two fit on a CU.)  Chain = N convolutions 3x3 C->C on a small map, each consuming the last.

    python tools/overlap_probe.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from cc_amd import ops  # noqa: E402


def chain(x, w, b, n):
    for _ in range(n):
        x = ops.conv2d(x, w, b, 1, 1, "relu")
    return x


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    dev = torch.device("cuda", 0)
    N = 24
    for (C, H, W) in ((256, 16, 52), (128, 32, 104), (512, 8, 26)):
        g = torch.Generator(device=dev).manual_seed(0)
        xs = [torch.randn(4, C, H, W, device=dev, generator=g) for _ in range(2)]
        ws = [torch.randn(C, C, 3, 3, device=dev, generator=g) * 0.02 for _ in range(2)]
        bs = [torch.zeros(C, device=dev) for _ in range(2)]
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

        def serial():
            with torch.no_grad():
                chain(xs[0], ws[0], bs[0], N)
                chain(xs[1], ws[1], bs[1], N)

        def two_streams():
            cur = torch.cuda.current_stream()
            s1.wait_stream(cur); s2.wait_stream(cur)
            with torch.no_grad():
                with torch.cuda.stream(s1):
                    chain(xs[0], ws[0], bs[0], N)
                with torch.cuda.stream(s2):
                    chain(xs[1], ws[1], bs[1], N)
            cur.wait_stream(s1); cur.wait_stream(s2)
        t_serial = timed(serial)
        t_two = timed(two_streams)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            serial(); two_streams()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g1, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.graph(g1):
            serial()
        with torch.cuda.graph(g2):
            two_streams()
        t_g1 = timed(g1.replay)
        t_g2 = timed(g2.replay)
        print("C%d %dx%d, 2 chains x %d convs: eager one stream %.3f ms | eager two streams %.3f | graph one branch %.3f | graph two branches %.3f"
              % (C, H, W, N, t_serial, t_two, t_g1, t_g2), flush=True)


if __name__ == "__main__":
    main()
