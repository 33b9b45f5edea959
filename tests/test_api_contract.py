"""The drop-in contract of SURVEY.md 8b on the Python module surface: state_dict keys/shapes, constructor signatures,
train()/eval() return conventions, the reference's assertion messages and argument errors, checkpoint layout.
(The oracle nets are pinned to the unmodified reference by tests/test_oracle_golden.py.)"""
import io

import pytest
import torch

from cc_amd import models, inverse_warp as IW, loss_functions as LF, synthetic as syn
from hipemu.emu import emulated_engine
from oracle import nets as ON


@pytest.fixture(scope="module", autouse=True)
def _emu():
    with emulated_engine():
        yield


NETS = [("DispResNet6", {}), ("DispNetS", {}), ("PoseNetB6", dict(nb_ref_imgs=4)), ("MaskNet6", dict(nb_ref_imgs=4, output_exp=True)),
        ("PoseExpNet", dict(nb_ref_imgs=4, output_exp=True)), ("Back2Future", dict(nlevels=6))]


@pytest.mark.parametrize("name,kw", NETS)
def test_state_dict_contract(name, kw):
    ours, ref = getattr(models, name)(**kw), getattr(ON, name)(**kw)
    a, b = ours.state_dict(), ref.state_dict()
    assert list(a.keys()) == list(b.keys())
    assert [tuple(v.shape) for v in a.values()] == [tuple(v.shape) for v in b.values()]
    # utils.py:55-63 checkpoint layout {'epoch', 'state_dict'} round-trips through torch.save / load_state_dict
    buf = io.BytesIO()
    torch.save({"epoch": 3, "state_dict": b}, buf)
    buf.seek(0)
    ck = torch.load(buf)
    ours.load_state_dict(ck["state_dict"])
    assert ck["epoch"] == 3


def test_train_eval_return_conventions():
    B, H, W = 2, 64, 64
    tgt, refs, K, Kinv = syn.sample(B, H, W, seed=1, smooth=2)
    disp, pose = models.DispResNet6(), models.PoseNetB6(nb_ref_imgs=4)
    disp.init_weights(), pose.init_weights()
    with torch.no_grad():
        out = disp.train()(tgt)
        assert isinstance(out, tuple) and len(out) == 6 and [tuple(o.shape[2:]) for o in out][:3] == [(64, 64), (32, 32), (16, 16)]
        one = disp.eval()(tgt)
        assert torch.is_tensor(one) and one.shape == (B, 1, H, W)                                   # DispResNet6.py:191-194
    assert pose.train()(tgt, refs).shape == (B, 4, 6)                                                # PoseNetB6.py:83
    m = models.MaskNet6(nb_ref_imgs=4, output_exp=True)
    m.init_weights()
    masks = m.train()(tgt, refs)
    assert len(masks) == 6 and masks[0].shape == (B, 4, H, W)


def test_reference_assertion_messages():
    img, depth = torch.zeros(2, 3, 8, 10), torch.ones(2, 8, 10)
    pose, K = torch.zeros(2, 6), torch.eye(3).repeat(2, 1, 1)
    with pytest.raises(AssertionError, match=r"wrong size for depth, expected BxHxW, got  \[2, 1, 8, 10\]"):
        IW.inverse_warp(img, depth.unsqueeze(1), pose, K, K)                                        # inverse_warp.py:23-28,263
    with pytest.raises(AssertionError, match=r"wrong size for pose, expected Bx6"):
        IW.inverse_warp(img, depth, torch.zeros(2, 7), K, K)
    with pytest.raises(AssertionError, match=r"wrong size for flow, expected Bx2xHxW"):
        IW.flow_warp(img, torch.zeros(2, 3, 8, 10))
    with pytest.raises(AssertionError, match=r"wrong size for flow, expected Bx2xHxW"):
        IW.flow2oob(torch.zeros(2, 8, 10))


def test_argument_errors_of_the_losses():
    B, H, W = 2, 16, 24
    tgt, refs, K, Kinv = syn.sample(B, H, W, seed=1)
    depth = [torch.ones(B, 1, H, W)]
    with pytest.raises(IndexError):           # depth_occlusion_masks indexes 4 reference poses (loss_functions.py:132-137)
        LF.photometric_reconstruction_loss(tgt, refs[:2], K, Kinv, depth, [None], torch.zeros(B, 2, 6))
    with pytest.raises(IndexError):           # occlusion_masks(flows[0], flows[1]) needs two flows (:70)
        LF.photometric_flow_loss(tgt, refs[:1], [[torch.zeros(B, 2, H, W)]], [None])
    with pytest.raises(AssertionError, match="wrong size for depth"):
        # B = 1: depth.squeeze() drops the batch dimension as well (loss_functions.py:133)
        LF.depth_occlusion_masks(torch.ones(1, 1, H, W), torch.zeros(1, 4, 6), K[:1], Kinv[:1])


def test_flow2oob_matches_definition():
    flow = torch.zeros(1, 2, 4, 6)
    flow[0, 0, 1, 5] = 0.5          # x + u = 5.5 > w - 1
    flow[0, 1, 0, 2] = -0.25        # y + v < 0
    oob = IW.flow2oob(flow)
    assert oob.dtype == torch.bool and oob.shape == (1, 4, 6)
    assert bool(oob[0, 1, 5]) and bool(oob[0, 0, 2]) and int(oob.sum()) == 2


def test_checkpoint_wire_format(tmp_path):
    """utils.py:55-63 + train.py:286-295,396-413: five files, {'epoch','state_dict'}, *_model_best copies; the files load
    with the reference's own torch.load + load_state_dict lines (here into the oracle nets, which carry the reference's
    keys) and back; the optimizer file has torch.optim.Adam's layout."""
    import os
    from cc_amd import trainer as T, utils
    torch.manual_seed(0)
    nets = T.build_nets("cpu")
    tr = T.CCTrainer(nets, T.StepConfig(), use_graph=False)
    tr.opt.exp_avg.normal_()                 # stand-in for the moments after one optimizer step
    tr.opt.exp_avg_sq.uniform_()
    tr.opt.step_dev.fill_(1.0)
    tr.save_checkpoint(tmp_path, epoch=4, is_best=True)
    names = sorted(os.listdir(tmp_path))
    assert names == sorted(["%s_%s" % (p, s) for p in utils.FILE_PREFIXES for s in ("checkpoint.pth.tar", "model_best.pth.tar")])
    # the reference's resume lines (train.py:288-295) against nets that carry the reference's keys
    ref_nets = [ON.DispResNet6(), ON.PoseNetB6(nb_ref_imgs=4), ON.MaskNet6(nb_ref_imgs=4, output_exp=True), ON.Back2Future(nlevels=6)]
    for prefix, net, ours in zip(utils.FILE_PREFIXES, ref_nets, nets):
        w = torch.load(os.path.join(tmp_path, prefix + "_checkpoint.pth.tar"))
        assert set(w.keys()) == {"epoch", "state_dict"} and w["epoch"] == 5
        net.load_state_dict(w["state_dict"])
        for (k, a), b in zip(net.state_dict().items(), ours.state_dict().values()):
            assert torch.equal(a, b), k
    # optimizer file: loadable by a real torch.optim.Adam over the same parameter chain
    params = [p for n in ref_nets for p in n.parameters()]
    opt = torch.optim.Adam(params, lr=1e-4, betas=(0.9, 0.999))
    osd = torch.load(os.path.join(tmp_path, "optimizer_checkpoint.pth.tar"))["state_dict"]
    opt.load_state_dict(osd)
    assert float(opt.state[params[0]]["step"]) == 1.0 and opt.state[params[0]]["exp_avg"].shape == params[0].shape
    # and back into a fresh engine trainer
    nets2 = T.build_nets("cpu")
    assert utils.resume(tmp_path, *nets2) == 5
    tr2 = T.CCTrainer(nets2, T.StepConfig(), use_graph=False)
    tr2.opt.load_state_dict(opt.state_dict())
    # (gather: the parameters' elements of the bucket, without the zero padding between the networks' 256-byte aligned ranges --
    # normal_() above filled the padding too, and a checkpoint carries parameters only)
    G, G2 = tr.opt.gather, tr2.opt.gather
    assert torch.equal(G2(tr2.opt.exp_avg), G(tr.opt.exp_avg)) and torch.equal(G2(tr2.opt.flat_p), G(tr.opt.flat_p))
    # train.py:311-314: the optimizer file is picked up on resume when it exists
    tr3 = T.CCTrainer(T.build_nets("cpu"), T.StepConfig(), use_graph=False)
    assert utils.resume_optimizer(tmp_path, tr3) is True
    assert torch.equal(tr3.opt.gather(tr3.opt.exp_avg), G(tr.opt.exp_avg)) and torch.equal(tr3.opt.gather(tr3.opt.exp_avg_sq), G(tr.opt.exp_avg_sq))
    assert float(tr3.opt.step_dev) == 1.0
    os.remove(os.path.join(tmp_path, "optimizer_checkpoint.pth.tar"))
    assert utils.resume_optimizer(tmp_path, tr3) is False


def test_nan_flags_are_never_dropped_unchecked():
    """a caller that checks rarely still sees an early NaN -- flags leaving the bounded list are OR-ed into a sticky
    flag instead of being deleted (the reference asserts on every term, loss_functions.py:60,105,115)."""
    from cc_amd import loss_functions as LF
    LF._nan_flags.clear()
    LF._sticky_nan.clear()
    LF._register_nan_flag(torch.ones(1))
    for _ in range(300):
        LF._register_nan_flag(torch.zeros(1))
    assert len(LF._nan_flags) <= 97
    with pytest.raises(AssertionError):
        LF.check_finite()
    LF.check_finite()                      # the failed check consumed the flags


def test_engine_rejects_strided_views_where_no_batch_stride_is_passed():
    """channel slices are read in place only by entry points that take a batch stride."""
    from cc_amd import _lib
    e = _lib.Engine(require_device=False)
    wide = torch.zeros(2, 8, 4, 6)
    sl = wide[:, 2:5]
    assert e._ptr(sl, "cc_conv2d_fwd", 0) == sl.data_ptr()
    with pytest.raises(ValueError):
        e._ptr(sl, "cc_ssim_fwd", 0)
