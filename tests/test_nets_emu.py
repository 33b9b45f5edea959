"""The four networks on the x86 emulation build of the kernels (the tape of cc_amd/tape.py: one autograd node per network,
hand-scheduled backward) against the oracle's torch restatement: outputs and ALL parameter gradients, at a size that
exercises the decoder crops (maps down to 1x1) -- the CPU-side twin of tests/test_nets_gpu.py::test_net_forward_backward."""
import pytest
import torch

from cc_amd import models, synthetic as syn
from hipemu.emu import emulated_engine
from oracle import nets as N


def _flat(o):
    if torch.is_tensor(o):
        return [o]
    r = []
    for x in o:
        if x is not None:
            r += _flat(x)
    return r


def _rel(a, b):
    a, b = a.detach().double(), b.detach().double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


CASES = {"disp": (models.DispResNet6, N.DispResNet6, (), "t"), "pose": (models.PoseNetB6, N.PoseNetB6, (4,), "tr"),
         "mask": (models.MaskNet6, N.MaskNet6, (4,), "tr"), "flow": (models.Back2Future, N.Back2Future, (6,), "t2")}


def _run(name, B, H, W, input_grad=False):
    tgt, refs, K, Kinv = syn.sample(B, H, W, seed=1)
    mk = CASES[name]
    mine, orc = mk[0](*mk[2]), mk[1](*mk[2])
    sd = syn.seeded_state_dict(orc, 0)
    mine.load_state_dict(sd)
    orc.load_state_dict(sd)
    t1, t0 = tgt.clone().requires_grad_(input_grad), tgt.clone().requires_grad_(input_grad)
    a1 = {"t": (t1,), "tr": (t1, refs), "t2": (t1, refs[1:3])}[mk[3]]
    a0 = {"t": (t0,), "tr": (t0, refs), "t2": (t0, refs[1:3])}[mk[3]]
    o1, o0 = _flat(mine(*a1)), _flat(orc(*a0))
    assert len(o1) == len(o0)
    for a, b in zip(o1, o0):
        assert a.shape == b.shape and _rel(a, b) < 1e-4, (name, tuple(a.shape), _rel(a, b))
    gen = torch.Generator().manual_seed(3)
    go = [torch.randn(a.shape, generator=gen) for a in o0]
    sum((a * g).sum() for a, g in zip(o1, go)).backward()
    sum((a * g).sum() for a, g in zip(o0, go)).backward()
    worst = (0.0, "")
    for (n1, p1), (n2, p2) in zip(mine.named_parameters(), orc.named_parameters()):
        if p2.grad is None:
            assert p1.grad is None or float(p1.grad.abs().max()) == 0, n1
            continue
        assert p1.grad is not None, n1
        r = float((p1.grad.double() - p2.grad.double()).norm() / (p2.grad.double().norm() + 1e-30))
        worst = max(worst, (r, n1))
    assert worst[0] < 2e-3, worst
    if input_grad:
        assert _rel(t1.grad, t0.grad) < 1e-3
    # eval mode / no grad: the same outputs without a tape
    mine.eval(), orc.eval()
    with torch.no_grad():
        e1, e0 = _flat(mine(*a1)), _flat(orc(*a0))
    for a, b in zip(e1, e0):
        assert _rel(a, b) < 1e-4
    return worst


@pytest.mark.parametrize("name", ["disp", "pose", "mask", "flow"])
def test_net_on_tape_emulated(name):
    with emulated_engine():
        # MaskNet6 concatenates without cropping (MaskNet6.py:98-103): its input size must divide by 64
        H, W = (64, 128) if name in ("mask", "flow") else (64, 96)
        print(name, _run(name, 2, H, W, input_grad=(name == "disp")))
