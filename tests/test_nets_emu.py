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


def test_tape_conv_group_shared_input_and_fresh_bias_buffers():
    """ADVICE r3 (tape.py conv_group.bwd / _act_bias): (a) a group whose inputs are [x1, x2, x2] where x1 already holds a gradient
    contribution when the group's backward runs -- the second use of x2 accumulates onto the first and must not be pulled into the
    earlier launch of x1's epilogue form; (b) bias gradients of group members whose output gradients arrive with different batch
    strides (slices of two concat buffers -> per-member launches) without an optimizer bucket: the fresh gradient buffers must be
    WRITTEN, not accumulated into (fresh buffers are poisoned with NaN here)."""
    import torch.nn.functional as F
    from cc_amd import tape as TP
    with emulated_engine():
        gen = torch.Generator().manual_seed(5)

        def rn(*s):
            return torch.randn(*s, generator=gen)
        B, C, H, W, M = 2, 6, 8, 12, 5
        x1, x2 = rn(B, C, H, W), rn(B, C, H, W)
        ws = [rn(M, C, 3, 3) * 0.2 for _ in range(4)]
        bs = [rn(M) * 0.3 for _ in range(4)]
        w0, b0 = rn(M, C, 3, 3) * 0.2, rn(M) * 0.3
        leaves = [x1, x2] + ws + bs + [w0, b0]
        td = [t.clone().requires_grad_(True) for t in leaves]
        tc = [t.clone().requires_grad_(True) for t in leaves]

        def ref(x1, x2, w_a, w_b, w_c, w_d, b_a, b_b, b_c, b_d, w0, b0):
            ys = [F.leaky_relu(F.conv2d(x, w, b, 1, 1), 0.2) for x, w, b in ((x1, w_a, b_a), (x2, w_b, b_b), (x2, w_c, b_c), (x1, w_d, b_d))]
            h = F.relu(F.conv2d(x1, w0, b0, 1, 1))                     # created AFTER the group: its backward runs first
            return [torch.cat(ys[:3], 1), torch.cat([ys[3], h], 1)]
        outs_c = ref(*tc)
        go = [rn(*o.shape) for o in outs_c]
        g0 = torch.autograd.grad(sum((o * g).sum() for o, g in zip(outs_c, go)), tc)

        real_empty = torch.empty_like

        def poisoned(t, *a, **k):                                       # fresh parameter-gradient buffers start as NaN
            r = real_empty(t, *a, **k)
            if r.dtype.is_floating_point:
                r.fill_(float("nan"))
            return r

        def body(tape, x1, x2):
            p = td[2:]
            ca = tape.concat(B, [M, M, M], H, W, x1.t)                  # batch stride 3 M H W
            cb = tape.concat(B, [M, M], H, W, x1.t)                     # batch stride 2 M H W
            ys = tape.conv_group([x1, x2, x2, x1], p[0:4], p[4:8], 1, 1, act="lrelu", outs=[ca.slot(0), ca.slot(1), ca.slot(2), cb.slot(0)])
            h = tape.conv(x1, p[8], p[9], 1, 1, act="relu", out=cb.slot(1))
            for i in range(3):
                ca.put(i, ys[i])
            cb.put(0, ys[3])
            cb.put(1, h)
            return [ca.done(), cb.done()]
        torch.empty_like = poisoned
        try:
            outs_d = TP.run_network(body, td[:2], td[2:])
            for a, b in zip(outs_d, outs_c):
                assert _rel(a, b) < 2e-5
            g1 = torch.autograd.grad(sum((o * g).sum() for o, g in zip(outs_d, go)), td, allow_unused=True)
        finally:
            torch.empty_like = real_empty
        for i, (a, b) in enumerate(zip(g1, g0)):
            assert a is not None and bool(torch.isfinite(a).all()), i
            assert _rel(a, b) < 2e-5, (i, _rel(a, b))
