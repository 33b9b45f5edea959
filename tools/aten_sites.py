#!/usr/bin/env python
"""Which lines of cc_amd still launch ATen kernels inside the training step?

    python tools/aten_sites.py [out.txt]        (on the GPU box)
    python tools/aten_sites.py --cpu [out.txt]  (anywhere: the step on the x86 emulation build at 2 x 64 x 128; every ATen operator
                                                 that computes something -- views excluded -- by the cc_amd source line that called it,
                                                 from a TorchDispatchMode: this is the exact list, the GPU form adds durations)

One eager step (B=4, 832x256, full CC).  (1) torch.profiler: the device kernels that do not come from libccengine, by ATen
operator.  (2) the Python-level tensor operations of cc_amd (add / mul / copy_ / fill / cat / zeros ... called from the package's
own code, in the forward pass or inside the backward of its autograd nodes) by source line; what the profiler counts beyond
those is the autograd engine's own gradient accumulation (AccumulateGrad / input-buffer adds)."""
import collections
import os
import sys
import traceback

import torch
from torch.profiler import ProfilerActivity, profile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cc_amd import synthetic as syn, trainer as T      # noqa: E402



def cpu_sites(out):
    """--cpu: dispatch-level list on the emulation build."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from torch.utils._python_dispatch import TorchDispatchMode
    from hipemu.emu import emulated_engine
    free = ("view", "slice", "select", "as_strided", "detach", "alias", "expand", "unbind", "unsqueeze", "squeeze", "permute", "transpose",
            "t.default", "_unsafe_view", "reshape", "split", "narrow", "empty", "_local_scalar", "unfold", "chunk", "set_", "resize_",
            "lift_fresh", "_to_copy")
    sites = collections.Counter()

    class Mode(TorchDispatchMode):
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            name = str(func)
            if not any(f in name for f in free):
                s = None
                for fr in reversed(traceback.extract_stack()[:-1]):
                    if "/cc_amd/" in fr.filename:
                        s = "%s:%d %s" % (os.path.relpath(fr.filename, ROOT), fr.lineno, (fr.line or "").strip()[:100])
                        break
                sites[(name.replace("aten.", ""), s or "(no cc_amd frame)")] += 1
            return func(*args, **(kwargs or {}))
    batch = syn.sample(2, 64, 128, seed=1)
    with emulated_engine():
        nets = T.build_nets("cpu")
        tr = T.CCTrainer(nets, T.StepConfig(), use_graph=False)
        tr.step(batch)
        with Mode():
            tr.step(batch)
    print("ATen operators that compute something in one training step (full CC; emulation build, 2 x 64 x 128), by calling line of cc_amd:", file=out)
    for (nm, s), n in sorted(sites.items(), key=lambda kv: (-kv[1], kv[0])):
        print("%4d  %-28s %s" % (n, nm, s), file=out)
    print("%4d  in total (the engine's own launches are not ATen operators and are not listed)" % sum(sites.values()), file=out)


if "--cpu" in sys.argv:
    args = [a for a in sys.argv[1:] if a != "--cpu"]
    cpu_sites(open(args[0], "w") if args else sys.stdout)
    sys.exit(0)

dev = torch.device("cuda", 0)
torch.manual_seed(0)
nets = T.build_nets(dev, flow=True, mask=True)
cfg = T.StepConfig()
b = syn.sample(4, 256, 832, seed=1, smooth=3)
batch = (b[0].to(dev), [r.to(dev) for r in b[1]], b[2].to(dev), b[3].to(dev))
tr = T.CCTrainer(nets, cfg, use_graph=False)
for _ in range(2):
    tr.step(batch)
torch.cuda.synchronize()
out = open(sys.argv[1], "w") if len(sys.argv) > 1 else sys.stdout

with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    tr.step(batch)
    torch.cuda.synchronize()
ops = collections.defaultdict(lambda: [0, 0.0])
for e in prof.events():
    if e.device_type != torch.autograd.DeviceType.CPU:
        continue
    ks = list(getattr(e, "kernels", []) or [])
    if not ks or all("k_" in k.name.split("(")[0] or "ccl" in k.name.lower() for k in ks):
        continue
    ops[e.name][0] += len(ks)
    ops[e.name][1] += sum(k.duration for k in ks)
print("non-engine device kernels in one eager step: %d launches, %.1f us" % (sum(v[0] for v in ops.values()), sum(v[1] for v in ops.values())), file=out)
for k, (n, t) in sorted(ops.items(), key=lambda kv: -kv[1][1]):
    print("%4d %8.1f us  %s" % (n, t, k), file=out)

# (2) Python-level call sites
sites = collections.Counter()


def site():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if "/cc_amd/" in fr.filename:
            return "%s:%d %s" % (os.path.relpath(fr.filename, ROOT), fr.lineno, (fr.line or "").strip()[:90])
    return None


def wrap(owner, name):
    orig = getattr(owner, name)

    def f(*a, **k):
        s = site()
        if s is not None and any(isinstance(x, torch.Tensor) and x.is_cuda for x in list(a) + list(k.values())):
            sites[(name, s)] += 1
        elif s is not None and name in ("zeros", "zeros_like", "ones", "ones_like", "full", "empty_like"):
            sites[(name, s)] += 1
        return orig(*a, **k)
    setattr(owner, name, f)


for nm in ("add_", "add", "__add__", "__iadd__", "__radd__", "sub", "__sub__", "__rsub__", "mul", "mul_", "__mul__", "__rmul__", "__imul__",
           "div", "__truediv__", "copy_", "fill_", "zero_", "clone", "contiguous", "mean", "sum", "neg", "__neg__"):
    wrap(torch.Tensor, nm)
for nm in ("cat", "zeros", "zeros_like", "ones_like", "full", "add", "mul", "stack"):
    wrap(torch, nm)
tr.step(batch)
torch.cuda.synchronize()
print("\nPython-level tensor operations of cc_amd in one step (calls; not every call is a kernel: contiguous() of a contiguous tensor is free):", file=out)
for (nm, s), n in sorted(sites.items(), key=lambda kv: (-kv[1], kv[0])):
    print("%4d  %-12s %s" % (n, nm, s), file=out)
