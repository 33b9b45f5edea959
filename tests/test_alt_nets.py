"""Alternative architectures of train.py:84-91: DispNetS6, DispResNetS6, PoseNet6, MaskResNet6, FlowNetC6 on the
engine's kernels against fixtures the unmodified reference modules produced (tests/golden/altnets.npz, written by
oracle/make_golden.py `alt_nets_level`): state_dict contract, train-mode outputs, parameter-gradient norm."""
import os

import numpy as np
import pytest
import torch

from cc_amd import models, synthetic as syn
from oracle.make_golden import ALT_NETS, AB, AH, AW, alt_net_args


def _check(dev, gold, name, kw):
    tgt, refs, K, Kinv = syn.sample(AB, AH, AW, seed=1)
    net = getattr(models, name)(**kw)
    net.load_state_dict(syn.seeded_state_dict(net, 0))
    net.to(dev).train()
    t, r = tgt.to(dev), [x.to(dev) for x in refs]
    out = net(*alt_net_args(name, t, r))
    outs = list(out) if isinstance(out, (tuple, list)) else [out]
    for i, o in enumerate(outs):
        k = "%s.%d" % (name, i)
        flat = o.detach().cpu().reshape(-1)
        if k in gold:
            ref = torch.from_numpy(gold[k]).reshape(-1)
            assert float((flat - ref).abs().max() / (ref.abs().max() + 1e-30)) < 1e-4, k
        else:
            st = gold[k + ".stats"]
            assert abs(float(flat.double().sum()) - st[0]) <= 1e-4 * st[1], k
            ref = torch.from_numpy(gold[k + ".sample"])
            got = flat[:: max(1, flat.numel() // 2048)]
            assert float((got - ref).abs().max() / (ref.abs().max() + 1e-30)) < 1e-4, k
    sum((o * o).mean() for o in outs).backward()
    gn = sum(float(p.grad.double().pow(2).sum()) for p in net.parameters() if p.grad is not None) ** 0.5
    assert abs(gn - float(gold[name + ".gradnorm"])) <= 2e-3 * float(gold[name + ".gradnorm"]), (gn, float(gold[name + ".gradnorm"]))


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "altnets.npz"))


@pytest.mark.parametrize("name,kw", ALT_NETS)
def test_alt_net_emulated(gold, name, kw):
    from hipemu.emu import emulated_engine
    with emulated_engine():
        _check("cpu", gold, name, kw)


@pytest.mark.gpu
@pytest.mark.parametrize("name,kw", ALT_NETS)
def test_alt_net_gpu(gold, name, kw):
    _check("cuda", gold, name, kw)
