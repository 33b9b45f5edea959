"""Parity at the configurations the metric is quoted on (BASELINE.json configs[1], [2] and the per-GPU shape of [4]):
the engine's full training step on the MI355X against scalars the UNMODIFIED reference produced at the same sizes
(tests/golden/headline.npz, written by `python -m oracle.make_golden headline`): six losses within 1e-4 rel (the
north-star bar), per-network gradient norms within 1e-3, the loss after one Adam step within 1e-4 -- eager and under hipGraph
replay (train.py:454-509,566-568)."""
import os

import numpy as np
import pytest
import torch

from cc_amd import synthetic as syn, trainer as T
from oracle.make_golden import HEADLINE

pytestmark = pytest.mark.gpu


def _run(golden_dir, tag, full, B, H, W, use_graph):
    g = dict(np.load(os.path.join(golden_dir, "headline.npz")))
    dev = torch.device("cuda")
    bc = syn.sample(B, H, W, seed=1, smooth=3)
    batch = (bc[0].to(dev), [r.to(dev) for r in bc[1]], bc[2].to(dev), bc[3].to(dev))
    nets = T.build_nets(dev, flow=full, mask=full, init=False)
    for n in nets:
        if n is not None:
            n.load_state_dict(syn.seeded_state_dict(n, 0))
    tr = T.CCTrainer(nets, T.StepConfig(), use_graph=use_graph)
    got = {k: float(v) for k, v in tr.step(batch).items()}
    report = {}
    for k in sorted(got):
        want = float(g["%s.%s" % (tag, k)])
        report[k] = abs(got[k] - want) / abs(want)
        assert report[k] <= 1e-4, (tag, k, got[k], want)
    for name, v in tr.grad_norms().items():
        want = float(g["%s.gradnorm.%s" % (tag, name)])
        report["gradnorm." + name] = abs(float(v) - want) / want
        assert report["gradnorm." + name] <= 1e-3, (tag, name, float(v), want)
    got2 = float(tr.step(batch)["loss"])
    want2 = float(g["%s.loss_after_adam" % tag])
    report["loss_after_adam"] = abs(got2 - want2) / abs(want2)
    assert report["loss_after_adam"] <= 1e-4, (tag, got2, want2)
    tr.check_finite()
    print("headline parity %s graph=%s: %s" % (tag, use_graph, {k: float("%.2e" % v) for k, v in report.items()}))


@pytest.mark.parametrize("use_graph", [False, True])
@pytest.mark.parametrize("case", HEADLINE, ids=[c[0] for c in HEADLINE])
def test_step_at_benchmarked_size(golden_dir, case, use_graph):
    _run(golden_dir, *case, use_graph)
