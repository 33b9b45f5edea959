#!/usr/bin/env python
"""What does one asynchronous all-reduce of a 1-rank RCCL group cost the compute stream?

    python tools/comm_probe.py          (on the GPU box; 127.0.0.1 rendezvous, world size 1)

A loop of [~1 ms of device work on the compute stream] [all_reduce(async) of n floats] [work.wait()] [a short kernel], timed with
events over 200 iterations, for several n and against the same loop without the collective: separates the fixed cost of the
process group's stream / event hand-offs from anything proportional to the buffer."""
import os
import time

import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29544")
dist.init_process_group("nccl", rank=0, world_size=1)
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
a = torch.randn(4096, 4096, device=dev)
big = torch.zeros(80 << 20, device=dev)           # 320 MB
small = torch.zeros(1024, device=dev)


def work():
    return a @ a                                  # ~0.9 ms of matrix work


def loop(n, iters=200, collective=True, touch=False):
    buf = big[:n] if n > 1024 else small[:n]
    for _ in range(10):
        work()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        work()
        if touch:
            buf.add_(1.0)                         # the buffer is dirty in L2 when the collective starts (as gradients are)
        if collective:
            w = dist.all_reduce(buf, async_op=True)
            w.wait()
        small.add_(1.0)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


base = loop(1024, collective=False)
print("loop without collective: %.3f ms/iter" % base)
for n in (1024, 1 << 20, 16 << 20, 56 << 20, 74 << 20):
    t = loop(n)
    print("all_reduce of %7.1f MB: %.3f ms/iter  (+%.3f)" % (4 * n / 1e6, t, t - base))
bt = loop(56 << 20, collective=False, touch=True)
t = loop(56 << 20, touch=True)
print("with the buffer written just before (56 M floats): without %.3f, with %.3f  (+%.3f)" % (bt, t, t - bt))
dist.destroy_process_group()
