#!/usr/bin/env python
"""When does each network's branch of the replayed step start and end -- WITHOUT a profiler attached (rocprofv3 slows the graph's
submission down: 20.2 ms instead of 16.4, and shows Back2Future's branch alone on the chip for 2.8 ms).  One-thread stamp kernels
(tools/stamp.hip: the GPU's 100 MHz clock) are launched at the points of interest while the step is captured, so they are nodes of
the graph; after every replay the stamps are read back.  Prints medians over the replays, in microseconds from the step's first node.
    python tools/branch_timeline.py [replays]"""
import ctypes, os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import ab_env
ab_env.apply()
from cc_amd import synthetic as syn, trainer as T, tape

HERE = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(HERE, "_bin", "libstamp.so")
if not os.path.isfile(so):
    os.makedirs(os.path.dirname(so), exist_ok=True)
    os.system("/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -shared -fPIC %s -o %s" % (os.path.join(HERE, "stamp.hip"), so))
lib = ctypes.CDLL(so)
lib.stamp_launch.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
dev = torch.device("cuda", 0)
torch.manual_seed(0)
nets = T.build_nets(dev, flow=True, mask=True)
b = syn.sample(4, 256, 832, seed=1, smooth=3)
batch = (b[0].to(dev), [r.to(dev) for r in b[1]], b[2].to(dev), b[3].to(dev))
tr = T.CCTrainer(nets, T.StepConfig(), use_graph=True)
buf = torch.zeros(64, dtype=torch.int64, device=dev)
names = {}


def stamp(i, name):
    names[i] = name
    lib.stamp_launch(buf.data_ptr() + 8 * i, torch.cuda.current_stream().cuda_stream)


NET = ("disp", "pose", "mask", "flow")
idx = {id(n): k for k, n in enumerate(nets)}
for k, n in enumerate(nets):
    n.register_forward_pre_hook(lambda m, a, k=k: stamp(1 + 2 * k, "%s forward starts" % NET[k]))
    n.register_forward_hook(lambda m, a, o, k=k: stamp(2 + 2 * k, "%s forward ends" % NET[k]))
_bwd = tape._NetFn.backward


def bwd(ctx, *g):
    k = idx.get(id(getattr(ctx.body, "__self__", None)), 0)
    stamp(10 + 3 * k, "%s backward starts" % NET[k])
    tape_done = tape.NET_DONE

    def done(mod):
        stamp(11 + 3 * k, "%s backward ends (tail starts)" % NET[k])
        if tape_done is not None:
            tape_done(mod)
        stamp(12 + 3 * k, "%s tail ends (Adam segment, weight images)" % NET[k])
    tape.NET_DONE = done
    try:
        return _bwd(ctx, *g)
    finally:
        tape.NET_DONE = tape_done


tape._NetFn.backward = staticmethod(bwd)
_begin, _lg, _sync = tr._begin, tr._loss_grads, tr._sync_streams
tr._begin = lambda bt: (stamp(0, "step starts"), _begin(bt))[1]


def lg(bt):
    r = _lg(bt)
    stamp(30, "losses + their gradients done (backward passes are enqueued next)")
    return r


tr._loss_grads = lg
# inside the loss phase (the only phase one stream runs alone): the rigid photometric loss (the step's stream; the side stream's chain
# runs beside it), the join + consensus loss, the weighted total, then the backward pass of the loss terms
from cc_amd import loss_functions as LF


def _wrap(name, i, label_a, label_b):
    f = getattr(LF, name)

    def g(*a, **k):
        if label_a:
            stamp(i, label_a)
        r = f(*a, **k)
        stamp(i + 1, label_b)
        return r
    setattr(LF, name, g)


_wrap("rigid_flows_levels", 32, "loss phase: rigid flows start", "  rigid flows issued")
_wrap("photometric_reconstruction_loss", 34, "  rigid photometric loss starts", "  rigid photometric loss ends")
_wrap("consensus_depth_flow_mask", 36, "  side stream joined, consensus loss starts", "  consensus loss ends")
_wrap("weighted_total", 38, None, "  weighted total done (the terms' backward calls follow)")
tr._sync_streams = lambda *a: (_sync(*a), stamp(31, "streams joined: step ends"))[0]

for _ in range(4):
    tr.step(batch)
torch.cuda.synchronize()
assert tr.graph is not None, "the step was not captured"
rows = []
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ms = []
# Steady state: replays are launched back to back (as a training loop does -- the host submits step k + 1 while step k runs, so no
# branch waits for its first packet); the stamps of the LAST replay of each burst are read.  "sync" as the second argument
# synchronises after every replay instead (then each step starts on an empty queue: the branches start 0.9 / 1.5 ms apart).
burst = 1 if (len(sys.argv) > 2 and sys.argv[2] == "sync") else 8
for _ in range(n):
    for k in range(burst):
        if k == burst - 1:
            ev0.record()
        tr.step(batch)
    ev1.record()
    torch.cuda.synchronize()
    ms.append(ev0.elapsed_time(ev1))
    rows.append(buf.cpu().tolist())
print("replayed step: median %.3f ms by HIP events (%d bursts of %d replays, %d stamp nodes in the graph)"
      % (statistics.median(ms), n, burst, len(names)))
med = {i: statistics.median((r[i] - r[0]) / 100.0 for r in rows) for i in names}
for i, t in sorted(med.items(), key=lambda kv: kv[1]):
    print("%10.1f us  %s" % (t, names[i]))
