#!/usr/bin/env python
"""A/B of the hand-written conv kernels against torch/MIOpen on the REAL network shapes (B=4, 256x832): per
parameter gradient difference of every net, NaN localisation.  GPU only."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cc_amd import config, models, synthetic as syn


def flat(o):
    if torch.is_tensor(o):
        return [o]
    r = []
    for x in o:
        if x is not None:
            r += flat(x)
    return r


def run(net, args, backend):
    config.conv_backend = backend
    for p in net.parameters():
        p.grad = None
    outs = flat(net(*args))
    sum((o * o).mean() for o in outs).backward()
    return [o.detach().clone() for o in outs], {n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None}


def main():
    dev = "cuda"
    B, H, W = int(os.environ.get("AB_B", 4)), int(os.environ.get("AB_H", 256)), int(os.environ.get("AB_W", 832))
    tgt, refs, K, Kinv = syn.sample(B, H, W, seed=1, smooth=3, device=dev)
    torch.manual_seed(0)
    nets = [("disp", models.DispResNet6(), (tgt,)), ("pose", models.PoseNetB6(4), (tgt, refs)),
            ("mask", models.MaskNet6(4), (tgt, refs)), ("flow", models.Back2Future(6), (tgt, refs[1:3]))]
    for name, net, args in nets:
        net.init_weights()
        net.to(dev).train()
        o_h, g_h = run(net, args, "hip")
        o_m, g_m = run(net, args, "miopen")
        worst_o = max(float((a - b).abs().max() / (b.abs().max() + 1e-30)) for a, b in zip(o_h, o_m))
        nan_o = any(bool(torch.isnan(a).any()) for a in o_h)
        bad = []
        for n in g_m:
            a, b = g_h[n], g_m[n]
            r = float((a - b).abs().max() / (b.abs().max() + 1e-30))
            if not (r < 1e-3):
                bad.append((n, tuple(a.shape), r, bool(torch.isnan(a).any())))
        print("%s: out rel %.2e nan=%s; %d/%d params off" % (name, worst_o, nan_o, len(bad), len(g_m)))
        for b in bad[:12]:
            print("   ", b)


if __name__ == "__main__":
    main()
