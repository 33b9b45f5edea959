"""TEST INFRASTRUCTURE ONLY (oracle) -- the Competitive-Collaboration training
step on CPU: a restatement of the loop body ``train.py:445-568`` (logging
removed) over either the oracle's own restated modules (default; travels to the
GPU box, used as ``bench.py``'s ``cpu_baseline`` of kind "port") or the
reference's modules (``impl=ref_import.load()``; build container only, used to
pin the restatement and to write tests/golden/).
"""
import types

import torch

from . import geometry as _geo
from . import losses as _los
from . import nets as _nets


class StepConfig:
    """README.md:59-65 recipe + train.py:34-135 defaults."""

    def __init__(self, **kw):
        self.w1, self.w2, self.w3, self.w4, self.w5 = 1.0, 0.1, 0.1, 0.5, 0.3   # -pc -m -s -pf -c
        self.wssim, self.wrig, self.wbce = 0.997, 1.0, 0.5
        self.THRESH, self.qch, self.lambda_oob = 0.01, 0.5, 0.0
        self.smoothness_type = "edgeaware"
        self.lr, self.betas = 1e-4, (0.9, 0.999)
        for k, v in kw.items():
            assert hasattr(self, k), k
            setattr(self, k, v)


def oracle_impl(align_corners=False):
    """Namespace with the same attribute layout ref_import.load() returns."""
    ac = align_corners

    iw = types.SimpleNamespace(
        pose2flow=_geo.pose2flow,
        inverse_warp=lambda *a, **k: _geo.inverse_warp(*a, align_corners=ac, **k),
        flow_warp=lambda *a, **k: _geo.flow_warp(*a, align_corners=ac, **k))
    lf = types.SimpleNamespace(
        photometric_reconstruction_loss=lambda *a, **k: _los.photometric_reconstruction_loss(*a, align_corners=ac, **k),
        photometric_flow_loss=lambda *a, **k: _los.photometric_flow_loss(*a, align_corners=ac, **k),
        consensus_exp_masks=lambda *a, **k: _los.consensus_exp_masks(*a, align_corners=ac, **k),
        consensus_depth_flow_mask=_los.consensus_depth_flow_mask,
        explainability_loss=_los.explainability_loss,
        edge_aware_smoothness_loss=_los.edge_aware_smoothness_loss,
        smooth_loss=_los.smooth_loss)
    return types.SimpleNamespace(inverse_warp=iw, loss_functions=lf)


def build_nets(kind="oracle", impl=None, align_corners=False, flow=True, mask=True):
    """disp, pose, mask, flow nets in the order train.py:245-255 creates them."""
    if kind == "oracle":
        disp, pose = _nets.DispResNet6(), _nets.PoseNetB6(nb_ref_imgs=4)
        msk = _nets.MaskNet6(nb_ref_imgs=4, output_exp=True) if mask else None
        flo = _nets.Back2Future(nlevels=6, align_corners=align_corners) if flow else None
    else:
        disp, pose = impl.DispResNet6.DispResNet6(), impl.PoseNetB6.PoseNetB6(nb_ref_imgs=4)
        msk = impl.MaskNet6.MaskNet6(nb_ref_imgs=4, output_exp=True) if mask else None
        flo = impl.back2future.Model(nlevels=6) if flow else None
    return [n for n in (disp, pose, msk, flo)]


def cc_forward(nets, batch, cfg, impl=None, keep=False, tap=None):
    """train.py:454-509 for the full CC configuration (all four nets).

    batch = (tgt[B,3,H,W], [4 refs], K[B,3,3], Kinv[B,3,3]).  Returns a dict with
    'loss', 'loss_1'..'loss_5' and (keep=True) the intermediates the parity tests
    compare."""
    impl = impl or oracle_impl()
    iw, lf = impl.inverse_warp, impl.loss_functions
    disp_net, pose_net, mask_net, flow_net = nets
    tgt, refs, K, Kinv = batch
    # tap (cc_step_keep): every network output passes through tap() on its way INTO the losses -- an identity whose hook sees the
    # gradient of the loss path alone (a DispResNet6 / Back2Future level also feeds the next decoder level inside the network)
    tp = (lambda ts: [tap(t) for t in ts]) if tap is not None else (lambda ts: ts)
    disparities = tp(disp_net(tgt))                                                # :454
    depth = [1 / d for d in disparities]                                           # :458
    pose = tp([pose_net(tgt, refs)])[0]                                            # :459
    out = {}
    if mask_net is None or flow_net is None:
        # BASELINE config 2: DispResNet6 + PoseNetB6, no mask, photometric + edge-aware smoothness
        l1 = lf.photometric_reconstruction_loss(tgt, refs, K, Kinv, depth, [None] * len(depth), pose,
                                                lambda_oob=cfg.lambda_oob, qch=cfg.qch, wssim=cfg.wssim)
        l3 = lf.edge_aware_smoothness_loss(tgt, depth)
        out.update(loss_1=l1, loss_3=l3, loss=cfg.w1 * l1 + cfg.w3 * l3)
        if keep:
            out.update(disparities=disparities, pose=pose)
        return out
    exp_mask = tp(mask_net(tgt, refs))                                             # :460
    flow_fwd, flow_bwd, _ = flow_net(tgt, refs[1:3])                               # :463
    flow_fwd, flow_bwd = tp(flow_fwd), tp(flow_bwd)
    cam_fwd = [iw.pose2flow(d.squeeze(1), pose[:, 2], K, Kinv) for d in depth]     # :470
    cam_bwd = [iw.pose2flow(d.squeeze(1), pose[:, 1], K, Kinv) for d in depth]     # :471
    target = lf.consensus_exp_masks(cam_fwd, cam_bwd, flow_fwd, flow_bwd, tgt, refs[2], refs[1],
                                    wssim=cfg.wssim, wrig=cfg.wrig, ws=cfg.w3)     # :473
    rig_fwd = [(a - b).abs() for a, b in zip(cam_fwd, flow_fwd)]                   # :475
    rig_bwd = [(a - b).abs() for a, b in zip(cam_bwd, flow_bwd)]                   # :476
    flow_exp_mask = [1 - m[:, 1:3] for m in exp_mask]                              # :488
    l1 = lf.photometric_reconstruction_loss(tgt, refs, K, Kinv, depth, exp_mask, pose,
                                            lambda_oob=cfg.lambda_oob, qch=cfg.qch, wssim=cfg.wssim)   # :490
    l2 = lf.explainability_loss(exp_mask) if cfg.w2 > 0 else 0                     # :492-495
    if cfg.smoothness_type == "regular":                                           # :497-501
        l3 = lf.smooth_loss(depth) + lf.smooth_loss(flow_fwd) + lf.smooth_loss(flow_bwd) + lf.smooth_loss(exp_mask)
    else:
        l3 = lf.edge_aware_smoothness_loss(tgt, depth) + lf.edge_aware_smoothness_loss(tgt, flow_fwd)
        l3 = l3 + lf.edge_aware_smoothness_loss(tgt, flow_bwd) + lf.edge_aware_smoothness_loss(tgt, exp_mask)
    l4 = lf.photometric_flow_loss(tgt, refs[1:3], [flow_bwd, flow_fwd], flow_exp_mask,
                                  lambda_oob=cfg.lambda_oob, qch=cfg.qch, wssim=cfg.wssim)             # :503
    l5 = lf.consensus_depth_flow_mask(exp_mask, rig_bwd, rig_fwd, target, target,
                                      THRESH=cfg.THRESH, wbce=cfg.wbce)            # :506
    loss = cfg.w1 * l1 + cfg.w2 * l2 + cfg.w3 * l3 + cfg.w4 * l4 + cfg.w5 * l5     # :509
    out.update(loss=loss, loss_1=l1, loss_2=l2, loss_3=l3, loss_4=l4, loss_5=l5)
    if keep:
        out.update(disparities=disparities, pose=pose, exp_mask=exp_mask, flow_fwd=flow_fwd,
                   flow_bwd=flow_bwd, cam_fwd=cam_fwd, cam_bwd=cam_bwd, target=target)
    return out


def make_optimizer(nets, cfg):
    """train.py:307-310: one Adam over all nets' parameters."""
    params = [p for n in nets if n is not None for p in n.parameters()]
    return torch.optim.Adam(params, lr=cfg.lr, betas=cfg.betas, weight_decay=0)


def cc_step(nets, optimizer, batch, cfg, impl=None):
    """train.py:454-509 + :566-568."""
    for n in nets:
        if n is not None:
            n.train()
    out = cc_forward(nets, batch, cfg, impl)
    optimizer.zero_grad()
    out["loss"].backward()
    optimizer.step()
    return {k: float(v) for k, v in out.items() if k.startswith("loss")}


KEEP_ORDER = ("disparities", "pose", "exp_mask", "flow_fwd", "flow_bwd")


def cc_step_keep(nets, optimizer, batch, cfg, impl=None):
    """cc_step() that also returns d loss / d (network outputs): the gradients the loss path sends into the four networks, in the
    order KEEP_ORDER (lists flattened, scale 0 first) -- what bench.py feeds into the engine's network backward passes to tell a
    difference that arises INSIDE a network's backward from one that arrives with its output gradients.  -> (losses, [grad or None])"""
    for n in nets:
        if n is not None:
            n.train()
    taps = []

    def tap(t):
        c = t.clone()                   # identity node between the network and the losses
        taps.append(c)
        if c.requires_grad:
            c.retain_grad()
        return c
    out = cc_forward(nets, batch, cfg, impl, keep=True, tap=tap)
    optimizer.zero_grad()
    out["loss"].backward()
    grads = [(t.grad.detach().clone() if t.grad is not None else None) for t in taps]      # tap order = KEEP_ORDER
    optimizer.step()
    return {k: float(v.detach() if torch.is_tensor(v) else v) for k, v in out.items() if k.startswith("loss")}, grads
