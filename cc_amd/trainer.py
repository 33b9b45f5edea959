"""The Competitive-Collaboration training step on the engine -- the host-side mirror of the loop body
train.py:445-568 (logging removed), written against the SAME call surface train.py uses
(``models.*``, ``inverse_warp.pose2flow``, ``loss_functions.*``), so it doubles as the drop-in demonstration.

MI355X-first structure around that body:
  * one process per GPU; each rank runs nets AND losses on its own mini-batch shard (the reference computes
    every loss on GPU 0 under nn.DataParallel, train.py:300-303);
  * all parameters live in ONE flat fp32 bucket (views re-pointed into it), gradients in a second one:
    a single RCCL all-reduce per step over xGMI (297 MB for the four nets) and a single fused Adam launch;
  * forward + backward of a step are captured once into a hipGraph (static shapes, no host syncs on the path)
    and replayed, which removes the ~1.5 k kernel-launch and Python/autograd dispatch costs from the step.
"""
import os

import torch
import torch.distributed as dist

from . import config, models, ops
from . import loss_functions as LF
from ._lib import engine, STREAM
from .inverse_warp import pose2flow


class StepConfig:
    """README.md:59-65 recipe; names follow train.py's argparse (train.py:34-135)."""

    def __init__(self, **kw):
        self.w1, self.w2, self.w3, self.w4, self.w5 = 1.0, 0.1, 0.1, 0.5, 0.3     # -pc -m -s -pf -c
        self.wssim, self.wrig, self.wbce = 0.997, 1.0, 0.5
        self.THRESH, self.qch, self.lambda_oob = 0.01, 0.5, 0.0
        self.smoothness_type = "edgeaware"
        self.lr, self.betas = 1e-4, (0.9, 0.999)
        for k, v in kw.items():
            assert hasattr(self, k), k
            setattr(self, k, v)


def build_nets(device, flow=True, mask=True, init=True):
    """disp, pose, mask, flow in the order train.py:245-255 creates (and :262-284 initialises) them."""
    disp = models.DispResNet6()
    pose = models.PoseNetB6(nb_ref_imgs=4)
    msk = models.MaskNet6(nb_ref_imgs=4, output_exp=True) if mask else None
    flo = models.Back2Future(nlevels=6) if flow else None
    nets = [disp, pose, msk, flo]
    for n in nets:
        if n is not None:
            if init:
                n.init_weights()
            n.to(device)
    return nets


def cc_forward(nets, batch, cfg, keep=False):
    """train.py:454-509.  batch = (tgt, [4 refs], K, Kinv)."""
    disp_net, pose_net, mask_net, flow_net = nets
    tgt, refs, K, Kinv = batch
    disparities = disp_net(tgt)                                                        # :454
    depth = [1 / d for d in disparities]                                               # :458
    pose = pose_net(tgt, refs)                                                         # :459
    out = {}
    if mask_net is None or flow_net is None:                                           # BASELINE config 2
        l1 = LF.photometric_reconstruction_loss(tgt, refs, K, Kinv, depth, [None] * len(depth), pose,
                                                lambda_oob=cfg.lambda_oob, qch=cfg.qch, wssim=cfg.wssim)
        l3 = LF.edge_aware_smoothness_loss(tgt, depth)
        out.update(loss_1=l1, loss_3=l3, loss=cfg.w1 * l1 + cfg.w3 * l3)
        if keep:
            out.update(disparities=disparities, pose=pose)
        return out
    exp_mask = mask_net(tgt, refs)                                                     # :460
    flow_fwd, flow_bwd, _ = flow_net(tgt, refs[1:3])                                   # :463
    cam_fwd = [pose2flow(d.squeeze(1), pose[:, 2], K, Kinv) for d in depth]            # :470
    cam_bwd = [pose2flow(d.squeeze(1), pose[:, 1], K, Kinv) for d in depth]            # :471
    target = LF.consensus_exp_masks(cam_fwd, cam_bwd, flow_fwd, flow_bwd, tgt, refs[2], refs[1],
                                    wssim=cfg.wssim, wrig=cfg.wrig, ws=cfg.w3)         # :473
    rig_fwd = [(a - b).abs() for a, b in zip(cam_fwd, flow_fwd)]                       # :475
    rig_bwd = [(a - b).abs() for a, b in zip(cam_bwd, flow_bwd)]                       # :476
    flow_exp_mask = [1 - m[:, 1:3] for m in exp_mask]                                  # :488
    l1 = LF.photometric_reconstruction_loss(tgt, refs, K, Kinv, depth, exp_mask, pose,
                                            lambda_oob=cfg.lambda_oob, qch=cfg.qch, wssim=cfg.wssim)   # :490
    l2 = LF.explainability_loss(exp_mask) if cfg.w2 > 0 else 0                         # :492-495
    if cfg.smoothness_type == "regular":                                               # :497-501
        l3 = LF.smooth_loss(depth) + LF.smooth_loss(flow_fwd) + LF.smooth_loss(flow_bwd) + LF.smooth_loss(exp_mask)
    else:
        l3 = LF.edge_aware_smoothness_loss(tgt, depth) + LF.edge_aware_smoothness_loss(tgt, flow_fwd)
        l3 = l3 + LF.edge_aware_smoothness_loss(tgt, flow_bwd) + LF.edge_aware_smoothness_loss(tgt, exp_mask)
    l4 = LF.photometric_flow_loss(tgt, refs[1:3], [flow_bwd, flow_fwd], flow_exp_mask,
                                  lambda_oob=cfg.lambda_oob, qch=cfg.qch, wssim=cfg.wssim)             # :503
    l5 = LF.consensus_depth_flow_mask(exp_mask, rig_bwd, rig_fwd, target, target,
                                      THRESH=cfg.THRESH, wbce=cfg.wbce)                # :506
    loss = cfg.w1 * l1 + cfg.w2 * l2 + cfg.w3 * l3 + cfg.w4 * l4 + cfg.w5 * l5         # :509
    out.update(loss=loss, loss_1=l1, loss_2=l2, loss_3=l3, loss_4=l4, loss_5=l5)
    if keep:
        out.update(disparities=disparities, pose=pose, exp_mask=exp_mask, flow_fwd=flow_fwd, flow_bwd=flow_bwd,
                   cam_fwd=cam_fwd, cam_bwd=cam_bwd, target=target)
    return out


class FlatAdam:
    """train.py:307-310 ``torch.optim.Adam(chain(all params), lr, betas)`` as ONE flat bucket + ONE kernel."""

    def __init__(self, nets, cfg):
        params = [p for n in nets if n is not None for p in n.parameters() if p.requires_grad]
        self.params = params
        dev = params[0].device
        n = sum(p.numel() for p in params)
        pad = (-n) % 4
        self.n = n
        self.flat_p = torch.zeros(n + pad, device=dev, dtype=torch.float32)
        self.flat_g = torch.zeros(n + pad, device=dev, dtype=torch.float32)
        self.exp_avg = torch.zeros_like(self.flat_p)
        self.exp_avg_sq = torch.zeros_like(self.flat_p)
        self.step_dev = torch.zeros(1, device=dev, dtype=torch.float32)
        off = 0
        for p in params:
            k = p.numel()
            self.flat_p[off:off + k].copy_(p.data.reshape(-1))
            p.data = self.flat_p[off:off + k].view_as(p.data)
            p.grad = self.flat_g[off:off + k].view_as(p.data)
            off += k
        self.lr, self.betas = cfg.lr, cfg.betas
        # conv weights / biases: the wgrad and bias-gradient kernels accumulate straight into the flat bucket
        # (the trainer switches ops.grad_sinks to this table for the duration of its own forward+backward only)
        self.sinks = {p.data_ptr(): p.grad for p in params}

    def zero_grad(self):
        engine().call("cc_fill", self.flat_g, self.flat_g.numel(), 0.0, STREAM)

    def all_reduce(self):
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(self.flat_g)          # RCCL over xGMI: one collective per step
            return 1.0 / dist.get_world_size()
        return 1.0

    def step(self, grad_scale=1.0):
        engine().call("cc_adam_step", self.flat_p, self.flat_g, self.exp_avg, self.exp_avg_sq, self.step_dev, self.n,
                      float(self.lr), float(self.betas[0]), float(self.betas[1]), 1e-8, float(grad_scale), STREAM)

    def state_dict(self):
        """The layout of ``torch.optim.Adam.state_dict()`` (what train.py:408-410 stores in optimizer_checkpoint.pth.tar):
        per-parameter ``step`` / ``exp_avg`` / ``exp_avg_sq`` cut out of the flat buckets, one param group."""
        state, off = {}, 0
        for i, p in enumerate(self.params):
            k = p.numel()
            state[i] = {"step": self.step_dev.detach().clone().reshape(()),
                        "exp_avg": self.exp_avg[off:off + k].view_as(p).clone(),
                        "exp_avg_sq": self.exp_avg_sq[off:off + k].view_as(p).clone()}
            off += k
        group = {"lr": self.lr, "betas": tuple(self.betas), "eps": 1e-8, "weight_decay": 0, "amsgrad": False,
                 "params": list(range(len(self.params)))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        """Accepts a ``torch.optim.Adam`` state dict over the same parameter order (chain of the four nets)."""
        off = 0
        for i, p in enumerate(self.params):
            k = p.numel()
            st = sd["state"].get(i)
            if st is not None:
                self.exp_avg[off:off + k].copy_(st["exp_avg"].reshape(-1))
                self.exp_avg_sq[off:off + k].copy_(st["exp_avg_sq"].reshape(-1))
                self.step_dev.fill_(float(st["step"]))
            off += k
        g = sd["param_groups"][0]
        self.lr, self.betas = g["lr"], tuple(g["betas"])

    def broadcast_from_rank0(self):
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            dist.broadcast(self.flat_p, 0)


class CCTrainer:
    """One rank of the data-parallel CC training job."""

    def __init__(self, nets, cfg, use_graph=True):
        self.nets, self.cfg = nets, cfg
        for n in nets:
            if n is not None:
                n.train()                                                   # train.py:438-441
        self.opt = FlatAdam(nets, cfg)
        self.opt.broadcast_from_rank0()
        self.use_graph = use_graph
        self.graph = None
        self.static_batch = None
        self.losses = None

    def _fwd_bwd(self, batch):
        LF.pyramid_cache.clear()
        ops.packs.prepack_all()            # every conv layer's [tap][c][m] weight images, one launch (weights changed in Adam)
        self.opt.zero_grad()                                                # :566
        ops.grad_sinks = self.opt.sinks
        LF.scalar_pool.begin(batch[0].device)
        try:
            out = cc_forward(self.nets, batch, self.cfg)
            out["loss"].backward()                                          # :567
        finally:
            LF.scalar_pool.end()
            ops.grad_sinks = {}
            ops.packs.invalidate()
        return {k: v.detach() for k, v in out.items() if torch.is_tensor(v)}

    def _copy_in(self, batch):
        tgt, refs, K, Kinv = batch
        s_tgt, s_refs, s_K, s_Kinv = self.static_batch
        s_tgt.copy_(tgt)
        for a, b in zip(s_refs, refs):
            a.copy_(b)
        s_K.copy_(K)
        s_Kinv.copy_(Kinv)

    def capture(self, batch, warmup=2):
        """Warm up eagerly on a side stream, then capture forward+backward of one step into a hipGraph."""
        tgt, refs, K, Kinv = batch
        self.static_batch = (tgt.clone(), [r.clone() for r in refs], K.clone(), Kinv.clone())
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(warmup):
                self._fwd_bwd(self.static_batch)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        # thread_local: only the capturing thread's calls are policed -- a process-group watchdog thread (multi-GPU runs)
        # polling its events must not invalidate the capture
        with torch.cuda.graph(self.graph, capture_error_mode=os.environ.get("CC_CAPTURE_MODE", "thread_local")):
            self.losses = self._fwd_bwd(self.static_batch)
        LF.pyramid_cache.clear()

    def save_checkpoint(self, save_path, epoch, is_best=False):
        """train.py:396-413: the five ``{'epoch', 'state_dict'}`` files of utils.save_checkpoint."""
        from . import utils
        sd = [({"epoch": epoch + 1, "state_dict": n.state_dict()} if n is not None else {"epoch": epoch + 1, "state_dict": {}})
              for n in self.nets]
        utils.save_checkpoint(save_path, sd[0], sd[1], sd[2], sd[3], {"epoch": epoch + 1, "state_dict": self.opt.state_dict()},
                              is_best)

    def step(self, batch):
        """train.py:445-568 for one mini-batch: returns the (device) loss tensors of this step."""
        if self.use_graph:
            if self.graph is None:
                self.capture(batch)
            self._copy_in(batch)
            self.graph.replay()
            losses = self.losses
        else:
            losses = self._fwd_bwd(batch)
        scale = self.opt.all_reduce()
        self.opt.step(scale)                                                # :568
        return losses
