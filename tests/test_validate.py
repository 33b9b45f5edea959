"""Validation loops (reference train.py:588-777): cc_amd.validate against fixtures produced by the
UNMODIFIED reference function bodies (tests/golden/validate.npz, oracle/make_golden.py `validate_level`: the FunctionDef nodes
of train.py / logger.py lifted out with `ast` and executed on the reference's networks)."""
import os

import numpy as np
import pytest
import torch

from cc_amd import models, synthetic as syn, validate as V
from cc_amd.logger import AverageMeter
from oracle.make_golden import RIGIDITY_NAMES, rigidity_inputs, validate_args, validate_inputs, validate_net_tweak


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "validate.npz"))


def _rigidity(dev, gold):
    ri = rigidity_inputs()
    r = V.rigidity_composition(ri["explainability_mask"].to(dev), ri["flow_cam"].to(dev), ri["flow_fwd"].to(dev), ri["THRESH"])
    pairs = dict(oob_rigid=r.oob_rigid, oob_non_rigid=r.oob_non_rigid, rigidity_mask=r.rigidity_mask,
                 rigidity_mask_census=r.rigidity_mask_census, rigidity_mask_combined=r.rigidity_mask_combined,
                 flow_fwd_non_rigid=r.flow_fwd_non_rigid, flow_fwd_rigid=r.flow_fwd_rigid, total_flow=r.total_flow)
    for k, t in pairs.items():
        ref = gold["rigidity." + k]
        got = t.float().cpu().numpy()
        assert got.shape == ref.shape, (k, got.shape, ref.shape)
        assert np.array_equal(got, ref), (k, int((got != ref).sum()))          # masks and masked flows: bit-exact
    # every branch of the composition is populated in the fixture
    for k in ("rigidity_mask", "rigidity_mask_census", "rigidity_mask_combined", "oob_rigid"):
        m = gold["rigidity." + k].mean()
        assert 0.1 < m < 0.95, (k, m)
    only = V.rigidity_composition(ri["explainability_mask"].to(dev), ri["flow_cam"].to(dev), ri["flow_fwd"].to(dev), ri["THRESH"],
                                  want=("total_flow",))
    assert only.rigidity_mask is None and np.array_equal(only.total_flow.cpu().numpy(), gold["rigidity.total_flow"])


def _nets(dev):
    nets = [models.DispResNet6(), models.PoseNetB6(nb_ref_imgs=4), models.MaskNet6(nb_ref_imgs=4, output_exp=True),
            models.Back2Future(nlevels=6)]
    for n in nets:
        n.load_state_dict(syn.seeded_state_dict(n, 0))
    validate_net_tweak(nets[2])
    return [n.to(dev) for n in nets]


def _loops(dev, gold, tol, rerun=True):
    nets = _nets(dev)
    flow_items, depth_items = validate_inputs()
    args = validate_args()
    seen = []
    err, names = V.validate_flow_with_gt(flow_items, *nets, 0, None, args=args, on_sample=lambda i, d: seen.append(d))
    assert names == [str(n) for n in gold["flow.names"]]
    ref = gold["flow.errors"]
    assert np.max(np.abs(np.asarray(err) - ref) / np.maximum(np.abs(ref), 1e-3)) < tol, (err, ref)
    assert len(seen) == len(flow_items) and seen[0]["total_flow"].shape == seen[0]["flow_fwd"].shape
    assert all(not n.training for n in nets)
    if rerun:
        err2, _ = V.validate_flow_with_gt(flow_items, *nets, 0, None, args=args)      # without the hook: two outputs only
        assert np.allclose(err, err2, rtol=1e-6)
    err, names = V.validate_depth_with_gt(depth_items, nets[0], 0, None, args=args)
    assert names == [str(n) for n in gold["depth.names"]]
    ref = gold["depth.errors"]
    assert np.max(np.abs(np.asarray(err) - ref) / np.maximum(np.abs(ref), 1e-3)) < tol, (err, ref)
    args.spatial_normalize = True
    err, _ = V.validate_depth_with_gt(depth_items, nets[0], 0, None, args=args)
    ref = gold["depth.errors_spatial_normalize"]
    assert np.max(np.abs(np.asarray(err) - ref) / np.maximum(np.abs(ref), 1e-3)) < tol, (err, ref)


def test_rigidity_names_cover_reference_statements(gold):
    assert all(("rigidity." + k) in gold for k in RIGIDITY_NAMES)


def test_average_meter():
    m = AverageMeter(i=2)
    m.update([1.0, torch.tensor(2.0)])
    m.update([3.0, torch.tensor(6.0)], n=3)
    assert m.count == 4 and abs(m.avg[0] - 2.5) < 1e-12 and abs(float(m.avg[1]) - 5.0) < 1e-6
    assert repr(m) == "3.000 6.000 (2.500 5.000)"
    with pytest.raises(AssertionError):
        m.update([1.0])


def test_rigidity_composition_emulated(gold):
    from hipemu.emu import emulated_engine
    with emulated_engine():
        _rigidity("cpu", gold)


def test_validate_loops_emulated(gold):
    from hipemu.emu import emulated_engine
    with emulated_engine():
        _loops("cpu", gold, 2e-3, rerun=False)


@pytest.mark.gpu
def test_rigidity_composition_gpu(gold):
    _rigidity("cuda", gold)


@pytest.mark.gpu
def test_validate_loops_gpu(gold):
    _loops("cuda", gold, 2e-3)
