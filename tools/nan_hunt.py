#!/usr/bin/env python
"""Run a few eager CC steps on the bench configuration printing every loss term, for the hip and miopen conv backends."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cc_amd import config, synthetic as syn, trainer as T, loss_functions as LF

def run(backend, steps, use_graph):
    config.conv_backend = backend
    dev = torch.device("cuda")
    torch.manual_seed(0)
    nets = T.build_nets(dev)
    bc = syn.sample(4, 256, 832, seed=1, smooth=3)
    batch = (bc[0].to(dev), [r.to(dev) for r in bc[1]], bc[2].to(dev), bc[3].to(dev))
    tr = T.CCTrainer(nets, T.StepConfig(), use_graph=use_graph)
    for i in range(steps):
        l = tr.step(batch)
        gn = float(tr.opt.flat_g.norm())
        print(backend, "graph" if use_graph else "eager", i, " ".join("%s=%.5f" % (k, float(v)) for k, v in sorted(l.items())), "gradnorm %.4g" % gn, flush=True)

steps = int(os.environ.get("STEPS", 14))
run("hip", steps, False)
run("miopen", steps, False)
run("hip", steps, True)
