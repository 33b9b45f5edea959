#!/usr/bin/env python
"""Host time of one training step (how long the Python thread is busy enqueueing one hipGraph replay + the optimizer launch)
against the device time of the step: is the step close to host-bound?   python tools/host_probe.py   (CC_FORCE_COMM=1 for the
data-parallel step form on a one-rank RCCL group)"""
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cc_amd import synthetic as syn, trainer as T      # noqa: E402

from tools import ab_env                                # noqa: E402
comm = bool(ab_env.apply().get("force_comm"))
if comm:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29545")
    dist.init_process_group("nccl", rank=0, world_size=1)
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
torch.manual_seed(0)
nets = T.build_nets(dev, flow=True, mask=True)
b = syn.sample(4, 256, 832, seed=1, smooth=3)
batch = (b[0].to(dev), [r.to(dev) for r in b[1]], b[2].to(dev), b[3].to(dev))
tr = T.CCTrainer(nets, T.StepConfig(), use_graph=True)
for _ in range(5):
    tr.step(batch)
torch.cuda.synchronize()
N = 30
host = []
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
t_all = time.perf_counter()
for _ in range(N):
    t0 = time.perf_counter()
    tr.step(batch)
    host.append(time.perf_counter() - t0)
e1.record()
t_enq = time.perf_counter() - t_all
torch.cuda.synchronize()
dev_ms = e0.elapsed_time(e1) / N
host.sort()
print("%s: device %.3f ms/step; host busy per step: median %.3f ms, p90 %.3f ms, max %.3f ms; %d steps enqueued in %.1f ms"
      % ("comm" if comm else "plain", dev_ms, 1e3 * host[N // 2], 1e3 * host[int(0.9 * N)], 1e3 * host[-1], N, 1e3 * t_enq))
# the same with a synchronisation before every step: the host starts each step with an idle GPU (no enqueue-ahead)
torch.cuda.synchronize()
t = []
for _ in range(10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tr.step(batch)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t.append((t1 - t0, time.perf_counter() - t0))
t.sort()
print("   synchronised steps: host enqueue %.3f ms, until the device is done %.3f ms" % (1e3 * t[5][0], 1e3 * t[5][1]))
if comm:
    dist.destroy_process_group()
