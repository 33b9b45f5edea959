"""The Competitive-Collaboration training step on the engine -- the host-side mirror of the loop body
train.py:445-568 (logging removed), written against the SAME call surface train.py uses
(``models.*``, ``inverse_warp.pose2flow``, ``loss_functions.*``), so it doubles as the drop-in demonstration.

MI355X-first structure around that body:
  * one process per GPU; each rank runs nets AND losses on its own mini-batch shard (the reference computes
    every loss on GPU 0 under nn.DataParallel, train.py:300-303);
  * all parameters live in ONE flat fp32 bucket (views re-pointed into it), gradients in a second one:
    a single RCCL all-reduce per step over xGMI (297 MB for the four nets) and a single fused Adam launch;
  * forward + backward of a step are captured once into a hipGraph (static shapes, no host syncs on the path)
    and replayed, which removes the ~1.5 k kernel-launch and Python/autograd dispatch costs from the step;
  * round 6, pipeline "per_network" (the default): the four networks run on HIP streams of their own, and at the end of a
    network's backward pass ITS stream exchanges ITS segment of the gradient bucket (RCCL all-reduce enqueued on that stream:
    a node of the graph branch), runs ITS Adam segment and rebuilds ITS weight images -- under the other networks' backward
    passes.  Only the last finisher's exchange is exposed; on one GPU the optimizer and the weight-image refresh leave the
    critical path the same way.
"""
import contextlib

import torch
import torch.distributed as dist

from . import config, models, ops, tape
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


def _fork(streams):
    """the given side streams wait for the current stream (inside a hipGraph capture this is also what brings them into it)"""
    streams = [st for st in streams if st is not None]
    if not streams:
        return
    cur = torch.cuda.current_stream()
    for st in streams:
        if st is not None and st.cuda_stream != cur.cuda_stream:
            st.wait_stream(cur)


def _join(streams):
    """the current stream waits for the given side streams (each must have been forked from it in this capture region)"""
    streams = [st for st in streams if st is not None]
    if not streams:
        return
    cur = torch.cuda.current_stream()
    for st in streams:
        if st is not None and st.cuda_stream != cur.cuda_stream:
            cur.wait_stream(st)


def _tensors(x):
    if torch.is_tensor(x):
        yield x
    elif isinstance(x, (list, tuple)):
        for y in x:
            yield from _tensors(y)


def cc_forward(nets, batch, cfg, keep=False, cut=None, streams=None, after_fork=None):
    """train.py:454-509.  batch = (tgt, [4 refs], K, Kinv).
    cut: optional dict; when given, the losses are computed on detached copies of the network outputs and
    cut['dp'] / cut['mf'] receive the (output, detached copy) pairs of DispResNet6 + PoseNetB6 / MaskNet6 + Back2Future, so
    that the backward pass can be run in stages (losses -> copies, then each network group from its outputs)."""
    disp_net, pose_net, mask_net, flow_net = nets
    tgt, refs, K, Kinv = batch

    def _cut(group, ts):
        if cut is None:
            return ts
        outs = []
        for t in ts:
            if t.requires_grad:
                d = t.detach().requires_grad_(True)
                cut.setdefault(group, []).append((t, d))
                outs.append(d)
            else:
                outs.append(t)
        return outs

    forked = []

    def _on(st, fn):
        """fn() on side stream st (config.net_streams: streams = (for DispResNet6, for Back2Future)), forked from the current one"""
        if st is None:
            return fn()
        _fork([st])
        forked.append(st)
        with torch.cuda.stream(st):
            return fn()
    s_disp, s_flow, s_mask = (tuple(streams) + (None, None, None))[:3] if streams else (None, None, None)
    full = not (mask_net is None or flow_net is None)
    # (independent of each other between the frames and the losses: the side streams' networks are launched first so that they have
    # work while the step's stream runs the other two; train.py:454-463 order of the results is kept below.  The autograd engine
    # runs ready nodes latest-created first, so the backward passes are ENQUEUED pose, mask, disp, flow: shortest first, which is the
    # order the per-network pipeline issues the networks' gradient exchanges in -- one RCCL communicator executes its collectives in
    # issue order, and Back2Future, the longest backward, must not hold up DispResNet6's segment)
    flow_out = _on(s_flow, lambda: flow_net(tgt, refs[1:3])) if full else None         # :463
    disp_out = _on(s_disp, lambda: list(disp_net(tgt)))                                # :454
    # (behind the side streams' launches, on the step's stream: nothing of it is needed before the losses / the backward pass, and
    # in front of the forks it would hold up DispResNet6, whose chain is the step's critical path)
    if after_fork is not None:
        after_fork()
    LF.pyramid_cache.prefetch([tgt] + list(refs))        # the frames' scale pyramids (every loss pools them): one launch for all five
    mask_out = _on(s_mask, lambda: list(mask_net(tgt, refs))) if full else None        # :460
    pose_out = pose_net(tgt, refs)                                                     # :459
    _join(forked)
    if forked and not torch.cuda.is_current_stream_capturing():
        # eager mode: these tensors were allocated on a side stream and are read by the loss kernels on this one -- tell the caching
        # allocator, so that a block freed here is not handed back to its home stream while this stream still reads it (under
        # capture the graph's private pool is not recycled across streams: nothing to record)
        cur = torch.cuda.current_stream()
        for st, outs in ((s_disp, disp_out), (s_flow, flow_out), (s_mask, mask_out)):
            if st is not None and outs is not None:
                for t in _tensors(outs):
                    t.record_stream(cur)
    disparities = _cut("dp", disp_out)
    depth = LF.reciprocal_levels(disparities)                                          # :458  [1 / d for d in disparities]
    pose = _cut("dp", [pose_out])[0]
    out = {}
    if mask_net is None or flow_net is None:                                           # BASELINE config 2
        l1 = LF.photometric_reconstruction_loss(tgt, refs, K, Kinv, depth, [None] * len(depth), pose,
                                                lambda_oob=cfg.lambda_oob, qch=cfg.qch, wssim=cfg.wssim)
        l3 = LF.edge_aware_smoothness_loss(tgt, depth)
        out.update(loss_1=l1, loss_3=l3, loss=cfg.w1 * l1 + cfg.w3 * l3)
        if keep:
            out.update(disparities=disparities, pose=pose)
        return out
    exp_mask = _cut("mf", mask_out)
    flow_fwd, flow_bwd, _ = flow_out
    flow_fwd, flow_bwd = _cut("mf", list(flow_fwd)), _cut("mf", list(flow_bwd))
    cam_fwd, cam_bwd = LF.rigid_flows_levels(depth, pose, (2, 1), K, Kinv)             # :470-471  pose2flow per scale
    flow_exp_mask = LF.complement_slice_levels(exp_mask, 1, 3)                         # :488  1 - m[:, 1:3]
    # config.loss_stream: the loss phase is the one part of the step a single stream runs alone (every network's forward pass is done,
    # no backward pass can start: 1.2 ms of latency-bound kernels, tools/branch_timeline.py).  Its two independent halves -- the
    # consensus target + the flow photometric loss, and the rigid photometric loss + the mask / smoothness terms -- run side by side:
    # Back2Future's stream (idle between the passes) is forked here and joined before the consensus loss, which needs the target.
    # The order of the CALLS is unchanged, so the terms meet the shared gradient accumulators in the same order as on one stream.
    side = s_flow if (config.loss_stream and s_flow is not None) else None
    origin = torch.cuda.current_stream() if side is not None else None
    if side is not None:
        side.wait_stream(origin)
    with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):       # (no autograd node is created in here)
        target = LF.consensus_exp_masks(cam_fwd, cam_bwd, flow_fwd, flow_bwd, tgt, refs[2], refs[1],
                                        wssim=cfg.wssim, wrig=cfg.wrig, ws=cfg.w3)     # :473
        rig_fwd = LF.abs_diff_levels(cam_fwd, flow_fwd)                                # :475  (a - b).abs(), thresholded only
        rig_bwd = LF.abs_diff_levels(cam_bwd, flow_bwd)                                # :476
    l1 = LF.photometric_reconstruction_loss(tgt, refs, K, Kinv, depth, exp_mask, pose,
                                            lambda_oob=cfg.lambda_oob, qch=cfg.qch, wssim=cfg.wssim)   # :490
    LF.forward_stream = side         # (the forward launches of the terms below: on the side stream, behind the consensus target)
    try:
        l2 = LF.explainability_loss(exp_mask) if cfg.w2 > 0 else 0                     # :492-495
        if cfg.smoothness_type == "regular":                                           # :497-501
            l3s = [LF.smooth_loss(depth), LF.smooth_loss(flow_fwd), LF.smooth_loss(flow_bwd), LF.smooth_loss(exp_mask)]
        else:
            l3s = [LF.edge_aware_smoothness_sum(tgt, [depth, flow_fwd, flow_bwd, exp_mask])]  # the four terms, one job table
        l4 = LF.photometric_flow_loss(tgt, refs[1:3], [flow_bwd, flow_fwd], flow_exp_mask,
                                      lambda_oob=cfg.lambda_oob, qch=cfg.qch, wssim=cfg.wssim)         # :503
    finally:
        LF.forward_stream = None
    if side is not None:
        origin.wait_stream(side)
        if not torch.cuda.is_current_stream_capturing():      # (eager mode: allocated on the side stream, read on this one)
            for t in list(target) + list(rig_fwd) + list(rig_bwd) + [l4] + list(l3s) + ([l2] if torch.is_tensor(l2) else []):
                t.record_stream(origin)
    with torch.no_grad():
        l3 = l3s[0].detach() if len(l3s) == 1 else torch.stack(l3s).sum()              # reported; the total below takes the terms
    l5 = LF.consensus_depth_flow_mask(exp_mask, rig_bwd, rig_fwd, target, target,
                                      THRESH=cfg.THRESH, wbce=cfg.wbce)                # :506
    terms = [(cfg.w1, l1)] + ([(cfg.w2, l2)] if cfg.w2 > 0 else []) + [(cfg.w3, t) for t in l3s] + [(cfg.w4, l4), (cfg.w5, l5)]
    loss = LF.weighted_total([w for w, _ in terms], [t for _, t in terms])             # :509  w1*l1 + w2*l2 + w3*l3 + w4*l4 + w5*l5
    out.update(loss=loss, loss_1=l1, loss_2=l2, loss_3=l3, loss_4=l4, loss_5=l5)
    if keep:
        out.update(disparities=disparities, pose=pose, exp_mask=exp_mask, flow_fwd=flow_fwd, flow_bwd=flow_bwd,
                   cam_fwd=cam_fwd, cam_bwd=cam_bwd, target=target)
    return out


class _NoWork:
    def wait(self):
        pass


class _SideStreamWork:
    """measurement aid (CC_COMM_PROBE=sidestream): the stream / event hand-offs of an asynchronous collective -- compute stream ->
    side stream -> compute stream -- around a 16-byte kernel instead of the collective"""
    side = None

    def __init__(self, t):
        if _SideStreamWork.side is None:
            _SideStreamWork.side = torch.cuda.Stream()
        cur = torch.cuda.current_stream()
        self.side.wait_stream(cur)
        with torch.cuda.stream(self.side):
            t.add_(0.0)

    def wait(self):
        torch.cuda.current_stream().wait_stream(self.side)


class FlatAdam:
    """train.py:307-310 ``torch.optim.Adam(chain(all params), lr, betas)`` as ONE flat bucket + fused kernels.

    Bucket layout: the parameters in train.py:305's chain order (DispResNet6 | PoseNetB6 | MaskNet6 | Back2Future), every NETWORK's
    range starting on a 256-byte boundary (<= 63 zero floats of padding between two networks: zero gradient, zero update), so that
    each network is a segment that can be exchanged (RCCL), updated (cc_adam_step_segment, float4) and re-imaged on its own."""
    ALIGN = 64          # floats

    def __init__(self, nets, cfg):
        per_net = [([p for p in n.parameters() if p.requires_grad] if n is not None else []) for n in nets]
        params = [p for ps in per_net for p in ps]
        self.params = params
        # module buffers (BatchNorm running statistics / counters) and frozen parameters: not in the bucket, but part of what rank 0
        # hands to the other ranks at start-up (train.py:300-303: DataParallel replicates the whole module from device 0)
        self.extra_state = [b for n in nets if n is not None for b in n.buffers()] + \
                           [p for n in nets if n is not None for p in n.parameters() if not p.requires_grad]
        dev = params[0].device
        self.n = sum(p.numel() for p in params)          # parameters (without the alignment padding)
        self.net_ranges, self.offsets, off = [], [], 0   # per network: (lo, hi) of its parameters or None; per parameter: its offset
        for ps in per_net:
            if not ps:
                self.net_ranges.append(None)
                continue
            off = -(-off // self.ALIGN) * self.ALIGN
            lo = off
            for p in ps:
                self.offsets.append(off)
                off += p.numel()
            self.net_ranges.append((lo, off))
        size = -(-off // self.ALIGN) * self.ALIGN
        self.flat_p = torch.zeros(size, device=dev, dtype=torch.float32)
        self.flat_g = torch.zeros(size, device=dev, dtype=torch.float32)
        self.exp_avg = torch.zeros_like(self.flat_p)
        self.exp_avg_sq = torch.zeros_like(self.flat_p)
        self.step_dev = torch.zeros(1, device=dev, dtype=torch.float32)
        for p, off in zip(params, self.offsets):
            k = p.numel()
            self.flat_p[off:off + k].copy_(p.data.reshape(-1))
            p.data = self.flat_p[off:off + k].view_as(p.data)
            p.grad = self.flat_g[off:off + k].view_as(p.data)
        self.lr, self.betas = cfg.lr, cfg.betas
        # conv weights / biases: the wgrad and bias-gradient kernels accumulate straight into the flat bucket
        # (the trainer switches ops.grad_sinks to this table for the duration of its own forward+backward only)
        self.sinks = {p.data_ptr(): p.grad for p in params}
        ops.packs.reset()          # weight images registered against the pre-bucket storages are stale now
        # measurement aid (bench.py: what does the step cost without the exchange?): "skip" / "sidestream" REPLACE
        # the gradient exchange -- the ranks diverge.  Off unless a tools script sets it on the instance AND says so loudly.
        self.comm_probe = ""
        self._rccl = None
        self.n_comms = 1            # RCCL communicators of the captured form: one per issuing stream (rccl())

    def zero_grad(self):
        engine().call("cc_fill", self.flat_g, self.flat_g.numel(), 0.0, STREAM)

    def gather(self, flat):
        """the parameters' elements of a bucket-shaped tensor in chain order, without the alignment padding
        (== torch.cat([p.reshape(-1) for p in params]) for flat_p)"""
        return torch.cat([flat[lo:hi] for lo, hi in (r for r in self.net_ranges if r is not None)])

    def segment(self, i):
        """[lo, hi) of network i's bucket segment as exchanged / updated: from its first parameter to the next network's first
        (its padding included; the last one runs to the end of the bucket), or None"""
        r = self.net_ranges[i]
        if r is None:
            return None
        nxt = [q[0] for q in self.net_ranges[i + 1:] if q is not None]
        return r[0], (nxt[0] if nxt else self.flat_p.numel())

    @staticmethod
    def world():
        return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1

    @staticmethod
    def comm_active():
        """collectives are issued when there is more than one rank -- or when config.debug.force_comm asks for them on a one-rank
        process group (exercises the RCCL path on a single GPU: tests, tools)"""
        if not (dist.is_available() and dist.is_initialized()):
            return False
        return dist.get_world_size() > 1 or config.debug.force_comm

    def all_reduce(self, lo=0, hi=None, async_op=False):
        """SUM-all-reduce flat_g[lo:hi] over the ranks through torch.distributed (RCCL's process group on HIP devices, gloo in the
        CPU tests).  -> work handle or None."""
        if self.comm_active():
            hi = self.flat_g.numel() if hi is None else hi
            if hi > lo:
                if self.comm_probe == "skip":
                    return _NoWork()
                if self.comm_probe == "sidestream":
                    return _SideStreamWork(self.flat_g[lo:lo + 4])
                return dist.all_reduce(self.flat_g[lo:hi], async_op=async_op)
        return None

    def rccl(self):
        """the trainer's own RCCL communicators (cc_amd/rccl.py), created on first use OUTSIDE any stream capture: collectives that
        are enqueued on the caller's stream; HIP devices with an RCCL process group only, None otherwise (gloo: all_reduce()).
        ONE PER ISSUING STREAM (`n_comms`, set by CCTrainer: the step's own stream + its side streams): a communicator executes its
        collectives in issue order and RCCL orders launches of one communicator that come from different streams with edges of its
        own -- inside a capture those would be cross-stream edges in the middle of the graph (tools/capture_join_probe.py: 1-2.5 ms,
        and a side stream must never wait for a stream that waited for it).  A communicator that only ever sees one stream adds
        nothing to the graph but its kernel.  Every rank creates them in the same order."""
        if self._rccl is None and self.flat_g.is_cuda and self.comm_active() and dist.get_backend() == "nccl":
            from . import rccl
            self._rccl = tuple(rccl.Communicator(self.flat_g.device) for _ in range(max(1, int(self.n_comms))))
        return self._rccl

    def all_reduce_here(self, lo, hi, comm=0):
        """SUM-all-reduce flat_g[lo:hi] ORDERED ON THE CURRENT STREAM (what follows on this stream sees the sum; inside a capture it is
        a node of the graph): ncclAllReduce on this stream on HIP devices (communicator `comm`), a blocking process-group call
        otherwise."""
        if not self.comm_active() or hi <= lo or self.comm_probe == "skip":
            return
        comms = self.rccl()
        if comms is not None:
            comms[comm].all_reduce_sum_(self.flat_g[lo:hi])
        else:
            dist.all_reduce(self.flat_g[lo:hi])

    def grad_scale(self):
        return 1.0 / self.world()

    def step(self, grad_scale=1.0):
        engine().call("cc_adam_step", self.flat_p, self.flat_g, self.exp_avg, self.exp_avg_sq, self.step_dev, self.flat_p.numel(),
                      float(self.lr), float(self.betas[0]), float(self.betas[1]), 1e-8, float(grad_scale), STREAM)

    def tick(self):
        """advance the step counter alone (the segments of this step then update with tick = 0, from any stream)"""
        engine().call("cc_adam_tick", self.step_dev, STREAM)

    def step_segment(self, lo, hi, tick, grad_scale=1.0):
        """The update of elements [lo, hi) of the bucket (lo % 4 == 0); tick: advance the step counter (first segment only)."""
        hi = self.flat_p.numel() if hi is None else hi
        assert lo % 4 == 0 and 0 <= lo < hi <= self.flat_p.numel()
        engine().call("cc_adam_step_segment", self.flat_p[lo:hi], self.flat_g[lo:hi], self.exp_avg[lo:hi], self.exp_avg_sq[lo:hi],
                      self.step_dev, hi - lo, float(self.lr), float(self.betas[0]), float(self.betas[1]), 1e-8, float(grad_scale),
                      int(tick), STREAM)

    def state_dict(self):
        """The layout of ``torch.optim.Adam.state_dict()`` (what train.py:408-410 stores in optimizer_checkpoint.pth.tar):
        per-parameter ``step`` / ``exp_avg`` / ``exp_avg_sq`` cut out of the flat buckets, one param group."""
        state = {}
        for i, (p, off) in enumerate(zip(self.params, self.offsets)):
            k = p.numel()
            state[i] = {"step": self.step_dev.detach().clone().reshape(()),
                        "exp_avg": self.exp_avg[off:off + k].view_as(p).clone(),
                        "exp_avg_sq": self.exp_avg_sq[off:off + k].view_as(p).clone()}
        group = {"lr": self.lr, "betas": tuple(self.betas), "eps": 1e-8, "weight_decay": 0, "amsgrad": False,
                 "params": list(range(len(self.params)))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        """Accepts a ``torch.optim.Adam`` state dict over the same parameter order (chain of the four nets)."""
        for i, (p, off) in enumerate(zip(self.params, self.offsets)):
            k = p.numel()
            st = sd["state"].get(i)
            if st is not None:
                self.exp_avg[off:off + k].copy_(st["exp_avg"].reshape(-1))
                self.exp_avg_sq[off:off + k].copy_(st["exp_avg_sq"].reshape(-1))
                self.step_dev.fill_(float(st["step"]))
        g = sd["param_groups"][0]
        self.lr, self.betas = g["lr"], tuple(g["betas"])

    def broadcast_from_rank0(self):
        """Start-up: every rank continues from rank 0's parameters AND buffers (a --resume that only rank 0 read from disk, per-rank
        initialisation, a pretrained net loaded on one rank: train.py:257-295 run once, then DataParallel replicates)."""
        if self.comm_active():
            dist.broadcast(self.flat_p, 0)
            for t in self.extra_state:
                dist.broadcast(t.data, 0)


NET_NAMES = ("disp", "pose", "mask", "flow")


class CCTrainer:
    """One rank of the data-parallel CC training job."""

    def __init__(self, nets, cfg, use_graph=True, split_graphs=None, comm_debug=None, pipeline=None):
        """pipeline: how backward, gradient exchange, optimizer and weight-image refresh are arranged
             "per_network" (default)  each network's stream runs  flush -> all-reduce(its segment) -> Adam(its segment) -> weight images
                                      at the end of ITS backward pass, inside the one graph (eagerly on CPU tensors: same calls, same
                                      order -- the form the gloo tests run);
             "post"                   round 5: one graph, the [DispResNet6|PoseNetB6] and [MaskNet6|Back2Future] all-reduces issued
                                      behind it through the process group, segmented Adam, weight images at the next step's start;
             "staged"                 rounds 3-4: two graphs, the first segment's all-reduce between them (under backward stage B).
           split_graphs (legacy switch of bench.py / the tests): True = "staged", False = "post".
           comm_debug (measurement scripts only; default: plain product step): dict with any of
             'events': True  -- (post / staged) record HIP events around the waits for the gradient all-reduces (comm_stats())
             'join': 'single' -- (post / staged) one join + one optimizer launch instead of the segmented update (A/B)
             'probe': 'skip' | 'sidestream' -- REPLACE the all-reduces by nothing / a 16-byte side-stream kernel: the ranks diverge;
                      announced on stderr."""
        self.nets, self.cfg = nets, cfg
        self.comm_debug = dict(comm_debug or {})
        for n in nets:
            if n is not None:
                n.train()                                                   # train.py:438-441
        self.opt = FlatAdam(nets, cfg)
        self.opt.comm_probe = self.comm_debug.get("probe", "")
        if self.opt.comm_probe:
            import sys
            print("[ccengine] WARNING: comm_debug['probe'] = %r -- gradient all-reduces are NOT performed (measurement only)"
                  % self.opt.comm_probe, file=sys.stderr, flush=True)
        self.opt.broadcast_from_rank0()
        self.bn_counters = tape.BnCounters(nets)
        self.use_graph = use_graph
        self.graph = None
        self.graph_b = None
        # gradient segments of the legacy forms: [DispResNet6 | PoseNetB6] [MaskNet6 | Back2Future] (parameter order of
        # FlatAdam = train.py:305's chain); n_dp = first element of the second one
        tail = [r[0] for r in self.opt.net_ranges[2:] if r is not None]
        head = [r for r in self.opt.net_ranges[:2] if r is not None]
        self.n_dp = tail[0] if (tail and head) else (self.opt.flat_p.numel() if head else 0)
        two_segments = bool(tail and head)
        if pipeline is None:
            pipeline = "per_network" if split_graphs is None else ("staged" if split_graphs else "post")
        assert pipeline in ("per_network", "post", "staged"), pipeline
        if pipeline == "staged" and not two_segments:     # (MaskNet6 + Back2Future frozen -- README's --fix-masknet --fix-flownet)
            pipeline = "post"
        self.pipeline = pipeline
        self.split_graphs = pipeline == "staged"
        dev0 = next(p for n in nets if n is not None for p in n.parameters()).device
        # config.net_streams: one side stream each for DispResNet6 and Back2Future (forward AND backward: autograd runs a node's
        # backward on the stream of its forward); HIP devices only
        nst = int(config.net_streams) if config.net_streams else 0          # True / 2: two side streams; 3: a third one for MaskNet6
        nst = 2 if nst == 1 else nst
        # (stream priorities, config.debug.net_stream_priority: (DispResNet6's, Back2Future's[, MaskNet6's]); -1 = high.  Measured in
        # round 5, see profiles/r05_ab_round5.txt)
        pri = tuple(config.debug.net_stream_priority) + (0, 0, 0)
        self.net_streams = tuple(torch.cuda.Stream(dev0, priority=pri[i]) for i in range(nst)) if (nst and dev0.type == "cuda") else None
        self._net_index = {id(n): i for i, n in enumerate(nets) if n is not None}
        # gradient chunks (per_network): a network that marks points of its forward pass (module.GRAD_CHUNKS; DispResNet6, the last
        # finisher of the backward pass) hands the parameters behind a mark over as soon as its backward pass has come back to it:
        # (network index, tag) -> first element of the chunk in the bucket.  The chunks' tails run on the step's ORIGIN stream, which
        # has only PoseNetB6 + MaskNet6 to do and is idle long before DispResNet6 reaches its first mark (a FOURTH stream for them
        # costs the replayed graph +2.5 ms, profiles/r06_ab_round6.txt)
        self._chunk_lo, self._chunk_lo_all, self._open_hi, self._origin = {}, {}, {}, None
        self.grad_chunks = bool(config.grad_chunks)
        for i, n in enumerate(nets):        # (the table is built either way: set_grad_chunks switches the chunks on a live trainer)
            if n is None or self.opt.net_ranges[i] is None or not getattr(n, "GRAD_CHUNKS", None):
                continue
            off, at = self.opt.net_ranges[i][0], {}
            for name, p in n.named_parameters():
                if p.requires_grad:
                    at[name] = off
                    off += p.numel()
            for tag, first in n.GRAD_CHUNKS:
                if first in at and at[first] % 4 == 0:
                    self._chunk_lo_all[(i, tag)] = at[first]
        if pipeline == "per_network" and self.grad_chunks:
            self._chunk_lo = dict(self._chunk_lo_all)

        self._done = set()
        self.segment_calls = []          # per_network, most recent step: [(network index, lo, hi)] in issue order (tests, bench)
        self._packed_version = None      # flat_p._version the weight images were built for (per_network)
        self.comm_events = []            # per step: (before wait 0, after wait 0, before wait 1, after wait 1) on the compute stream
        self.comm_standalone_ms = None   # calibrate_comm(): each segment's all-reduce alone, nothing to hide under
        self.stage_b_events = []         # per step: events around the replay of backward stage B (two-graph form, comm_debug events)
        self.static_batch = None
        self.losses = None
        self.nan_flags = []
        self.opt.n_comms = 1 + (len(self.net_streams) if self.net_streams else 0)
        if self.pipeline == "per_network" and self.opt.comm_active():
            try:
                self.opt.rccl()          # the communicators must exist before any capture
            except Exception as e:       # noqa: BLE001 -- librccl not bindable / communicator set-up refused: keep training, say so
                self._fall_back("the direct RCCL communicators could not be created (%r)" % (e,))

    def switch_pipeline(self, pipeline):
        """Change the step form of a live trainer (same bucket, same optimizer state): the captured graphs are dropped and the next
        step captures the new form.  bench.py uses it at N > 1 to time the per-network form against round 5's on the machine at hand."""
        assert pipeline in ("per_network", "post", "staged"), pipeline
        if pipeline == "per_network" and self.opt.comm_active() and self.opt.flat_g.is_cuda:
            self.opt.rccl()
        self.pipeline, self.split_graphs = pipeline, pipeline == "staged"
        self.graph = self.graph_b = None
        self._chunk_lo = dict(self._chunk_lo_all) if (pipeline == "per_network" and self.grad_chunks) else {}
        self.comm_events, self.stage_b_events = [], []
        ops.packs.mark_stale()

    def set_grad_chunks(self, on):
        """config.grad_chunks for a live trainer (per-network form; the captured graph is dropped): bench.py times the form with and
        without the chunks at N > 1, where DispResNet6's exchange is what they start early."""
        self.grad_chunks = bool(on)
        self.switch_pipeline(self.pipeline)

    def _fall_back(self, why):
        """Data-parallel only: the per-network form needs ncclAllReduce on the networks' streams (cc_amd/rccl.py).  If that path is
        not available on a machine, the step falls back -- LOUDLY -- to round 5's form (one graph, the process group's two
        all-reduces behind it): slower, same results."""
        import sys
        print("[ccengine] WARNING: per-network gradient pipeline unavailable -- %s; falling back to pipeline='post' "
              "(process-group all-reduces behind the graph)" % why, file=sys.stderr, flush=True)
        self.pipeline, self.split_graphs = "post", False
        self.graph = self.graph_b = None
        self._chunk_lo = {}

    # ------------------------------------------------------------------------------------------------ the step's pieces
    def _begin(self, batch):
        tape.BN_COUNTERS = self.bn_counters
        self.bn_counters.begin()
        LF.pyramid_cache.clear()
        if self.pipeline == "per_network":
            ops.packs.begin_step()         # fresh from the previous step's per-network refresh: no launch
            self.opt.tick()                # the networks' Adam segments of this step all read the advanced counter
        else:
            ops.packs.prepack_all()        # every conv layer's [tap][c][m] weight images, one launch (weights changed in Adam)
        if not self.net_streams:
            self.opt.zero_grad()                                            # :566  (with side streams: behind their forks, cc_forward)
        ops.grad_sinks = self.opt.sinks
        LF.scalar_pool.begin(batch[0].device)

    def _loss_grads(self, batch):
        """forward of the four nets + losses (train.py:454-509) and d loss / d (network outputs) -> (losses, dp pairs, mf pairs)"""
        cut = {}
        LF.head_grads.begin()              # the loss terms' gradients of a shared network output meet in one accumulator
        try:
            out = cc_forward(self.nets, batch, self.cfg, cut=cut, streams=self.net_streams,
                             after_fork=self.opt.zero_grad if self.net_streams else None)
            self.bn_counters.commit()
            pairs = cut.get("dp", []) + cut.get("mf", [])
            g = torch.autograd.grad(out["loss"], [d for _, d in pairs], allow_unused=True) if pairs else ()   # :567, losses only
        finally:
            LF.head_grads.end()
        ndp = len(cut.get("dp", []))
        dp = [(t, gt) for (t, _), gt in zip(pairs[:ndp], g[:ndp]) if gt is not None]
        mf = [(t, gt) for (t, _), gt in zip(pairs[ndp:], g[ndp:]) if gt is not None]
        ops.wgrad_queue.enabled = not config.debug.no_wgrad_queue
        losses = {k: v.detach() for k, v in out.items() if torch.is_tensor(v) and k.startswith("loss")}
        return losses, dp, mf

    def _backward(self, pairs):
        """ONE backward call for the given (network output, gradient) pairs: every network's backward waits for the loss gradients
        only, so DispResNet6 (side stream 0), Back2Future (side stream 1) and PoseNetB6 + MaskNet6 (this stream) run side by side.
        (Two calls would order the second behind the join of the first.)  The gradient tensors stay referenced by the caller until
        the streams have been joined: they were allocated on this stream and are read on the others."""
        if not pairs:
            return
        self._origin = torch.cuda.current_stream() if self.net_streams else None
        _fork(self.net_streams or ())
        if self.net_streams and not torch.cuda.is_current_stream_capturing():
            for _, gt in pairs:                                  # (eager mode: see cc_forward)
                for st in self.net_streams:
                    gt.record_stream(st)
        torch.autograd.backward([t for t, _ in pairs], [gt for _, gt in pairs])

    # The legacy forms cut the step into stages so that the gradient exchange of the big DispResNet6 + PoseNetB6 segment (227 MB of the
    # 297 MB bucket) runs UNDER the backward pass of Back2Future + MaskNet6 (the longest stage):
    #   A: weight images, zero grads, forward of the four nets + losses (train.py:454-509), d loss / d (net outputs),
    #      backward of DispResNet6 + PoseNetB6                                   -> all-reduce(flat_g[:n_dp]) starts
    #   B: backward of MaskNet6 + Back2Future                                    -> all-reduce(flat_g[n_dp:]) (exposed)
    # The nets only meet in the losses, so the two backward stages are independent given the output gradients.
    def _stage_a(self, batch, fuse_b=False):
        self._begin(batch)
        losses, dp, mf = self._loss_grads(batch)
        if fuse_b and self.net_streams:
            self._backward(dp + mf)
            self._sync_streams(bool(dp + mf))
            mf = []
        else:
            self._backward(dp)
            self._sync_streams(bool(dp))     # the segment's gradients are complete before its all-reduce is issued
        return losses, mf

    def _sync_streams(self, forked=True):
        """join the networks' side streams (forked before the backward call), run what the backward stage has parked (each stream's
        own launches on that stream, re-forked from this one), join again: everything the stage produced is then ordered before
        what follows on this stream"""
        if self.net_streams and forked:
            _join(self.net_streams)
        sts = {st.cuda_stream: st for st in ops.wgrad_queue.streams() + ops.wgrad_reduces.streams()}
        _fork(sts.values())
        ops.wgrad_queue.flush()
        ops.wgrad_reduces.flush()
        _join(sts.values())

    def _stage_b(self, mf):
        self._backward(mf)
        self._sync_streams(bool(mf))

    def _stage_end(self, failed=False):
        tape.BN_COUNTERS = None
        tape.NET_DONE = tape.NET_MARK = None
        if failed:
            # a stage raised part-way: the parked launches' operands belong to the failed step and no fork / join is in place for
            # the side streams -- drop them instead of launching
            ops.wgrad_queue.drop()
            ops.wgrad_reduces.drop()
        else:
            ops.wgrad_queue.flush()
        ops.wgrad_queue.enabled = False
        LF.scalar_pool.end()
        ops.grad_sinks = {}
        if self.pipeline == "per_network" and not failed:
            ops.packs.end_step()
        else:
            ops.packs.invalidate()

    def _fwd_bwd(self, batch, between=None):
        """(legacy forms) forward + backward of one mini-batch; `between()` runs between the two backward stages."""
        ok = False
        try:
            losses, mf = self._stage_a(batch, fuse_b=between is None)      # (no collective between the stages: one backward call)
            if between is not None:
                between()
            self._stage_b(mf)
            ok = True
        finally:
            self._stage_end(failed=not ok)
        return losses

    # ------------------------------------------------------------------------------------------------ per-network pipeline
    def _net_done(self, module):
        """tape.NET_DONE: called at the end of `module`'s backward pass, inside its autograd node -- on the network's stream, behind
        its last gradient launch."""
        i = self._net_index.get(id(module))
        if i is not None and i not in self._done:
            self._finish_network(i)

    def _net_mark(self, module, tag):
        """tape.NET_MARK: the backward pass of `module` has come back to the point its forward marked `tag` -- the parameters from the
        chunk's first one up to the part already handed over are final (module.GRAD_CHUNKS).  Their tail starts now, on the
        network's TAIL stream, while the network's own stream goes on with the backward pass."""
        i = self._net_index.get(id(module))
        lo = self._chunk_lo.get((i, tag)) if i is not None else None
        if lo is None or i in self._done:
            return
        hi = self._open_hi.get(i, self.opt.segment(i)[1])
        if not (self.opt.segment(i)[0] < lo < hi):
            return
        self._open_hi[i] = lo
        cur, org = torch.cuda.current_stream() if self.net_streams else None, self._origin
        other = org if (org is not None and cur is not None and org.cuda_stream != cur.cuda_stream and not config.debug.chunk_inline) else None
        self._tail(i, lo, hi, other)

    def _tail(self, i, lo, hi, stream=None):
        """The tail of a gradient pipeline for elements [lo, hi) of network i's bucket segment: what the backward pass has parked on
        the CURRENT stream so far (weight-gradient groups, reduce and bias tables), then -- on `stream` if given (the step's origin
        stream, which waits for the current one; never the other way round: a side stream must not wait for a stream that waited
        for it, tools/capture_join_probe.py), else on the current stream -- the all-reduce of that range, its Adam update, its
        weight images."""
        ops.wgrad_queue._cur().flush()          # (this stream's queue; it flushes this stream's reduce / bias tables behind it)
        self.segment_calls.append((i, lo, hi))
        if NET_NAMES[i] in config.debug.pipe_skip_tail:      # (measurement: what does this network's tail cost the step?)
            return
        if stream is not None:
            stream.wait_stream(torch.cuda.current_stream())
        with (torch.cuda.stream(stream) if stream is not None else contextlib.nullcontext()):
            self.opt.all_reduce_here(lo, hi, comm=self._comm_index(stream))
            self.opt.step_segment(lo, hi, False, self.opt.grad_scale())
            for _ in range(config.debug.pipe_extra.get(NET_NAMES[i], 0)):        # (measurement: how much slack does this stream have?)
                engine().call("cc_fill", self._scratch(), self._scratch().numel(), 0.0, STREAM)
            base = self.opt.flat_p.data_ptr()
            ops.packs.repack_range(base + 4 * lo, base + 4 * hi)

    def _comm_index(self, stream=None):
        """the communicator of the stream a collective is issued on (`stream`, else the current one): 0 = the step's own stream,
        1 + k = side stream k -- the same on every rank, and never two streams on one communicator (FlatAdam.rccl)"""
        if not self.net_streams:
            return 0
        h = (stream if stream is not None else torch.cuda.current_stream()).cuda_stream
        for k, st in enumerate(self.net_streams):
            if st.cuda_stream == h:
                return 1 + k
        return 0

    def _scratch(self):
        if getattr(self, "_scr", None) is None:
            self._scr = torch.empty(64 << 20, device=self.opt.flat_p.device, dtype=torch.float32)      # 256 MB: ~50 us per fill
        return self._scr

    def _finish_network(self, i):
        """End of network i's backward pass (on its stream): the tail of what is left of its segment."""
        self._done.add(i)
        seg = self.opt.segment(i)
        if seg is None:
            return
        self._tail(i, seg[0], self._open_hi.get(i, seg[1]))

    def _step_pipelined(self, batch):
        """train.py:445-568 with every network's exchange + update behind ITS backward pass (see the class docstring)"""
        ok = False
        self._done = set()
        self._open_hi = {}
        self.segment_calls = []
        try:
            self._begin(batch)
            losses, dp, mf = self._loss_grads(batch)
            tape.NET_DONE, tape.NET_MARK = self._net_done, (self._net_mark if self._chunk_lo else None)
            both = dp + mf
            self._backward(both)
            tape.NET_DONE = tape.NET_MARK = None
            self._sync_streams(bool(both))
            for i in range(len(self.nets)):     # networks that are not on the tape (alternative architectures) or received no gradient
                if i not in self._done:
                    self._finish_network(i)
            ok = True
        finally:
            self._stage_end(failed=not ok)
        return losses

    def _weights_touched(self):
        """did anybody but this trainer write the parameters since the weight images were built?  (torch in-place ops bump the
        bucket's version counter -- load_state_dict, p.data.mul_(..); the Adam kernel does not)"""
        # (a Parameter re-pointed into the bucket keeps a version counter of its own: p.copy_() bumps p's, flat_p[..].copy_() the bucket's)
        v = self.opt.flat_p._version + sum(p._version for p in self.opt.params)
        if self._packed_version is None:
            self._packed_version = v
        if v != self._packed_version:
            self._packed_version = v
            return True
        return False

    def _copy_in(self, batch):
        tgt, refs, K, Kinv = batch
        s_tgt, s_refs, s_K, s_Kinv = self.static_batch
        # the captured graph has static shapes: a smaller last batch would silently broadcast into the buffers
        assert tgt.shape == s_tgt.shape and len(refs) == len(s_refs) and all(a.shape == b.shape for a, b in zip(refs, s_refs)) \
            and K.shape == s_K.shape and Kinv.shape == s_Kinv.shape, \
            "batch shapes differ from the captured step's (%s vs %s): use a fixed batch size / drop_last" % (
                tuple(tgt.shape), tuple(s_tgt.shape))
        dst, src = [s_tgt] + list(s_refs) + [s_K, s_Kinv], [tgt] + list(refs) + [K, Kinv]
        if all(t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() for t in src):
            torch._foreach_copy_(dst, src)          # one multi-tensor launch instead of seven copies in front of every replay
        else:
            for a, b in zip(dst, src):
                a.copy_(b)

    def capture(self, batch, warmup=2):
        """Warm up eagerly on a side stream, then capture forward+backward of one step into a hipGraph."""
        assert not config.strict_nan_checks, "config.strict_nan_checks syncs the host per loss term: use use_graph=False with it"
        tgt, refs, K, Kinv = batch
        self.static_batch = (tgt.clone(), [r.clone() for r in refs], K.clone(), Kinv.clone())
        # the eager warm-up passes must leave no trace: BatchNorm running statistics / num_batches_tracked would otherwise
        # absorb the first batch three times where the reference absorbs it once (train.py:454 runs each batch once) -- and the
        # pipelined step contains the optimizer, so parameters, moments and step counter are put back as well
        bn_state = [(b, b.detach().clone()) for n in self.nets if n is not None for b in n.buffers()]
        pipelined = self.pipeline == "per_network"
        opt_state = [(t, t.detach().clone()) for t in (self.opt.flat_p, self.opt.exp_avg, self.opt.exp_avg_sq, self.opt.step_dev)] \
            if pipelined else []
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(warmup):
                self._step_pipelined(self.static_batch) if pipelined else self._fwd_bwd(self.static_batch)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        for b, saved in bn_state + opt_state:
            b.copy_(saved)
        if pipelined:
            ops.packs.mark_stale()
            ops.packs.prepack_all()         # the images of the restored weights: the captured step starts from them
            ops.packs.end_step()
            self._packed_version = None
            self._weights_touched()
        LF.check_finite()                   # the warm-up's own flags (and drop them: the captured step registers its own)
        # thread_local: only the capturing thread's calls are policed -- a process-group watchdog thread (multi-GPU runs)
        # polling its events must not invalidate the capture
        mode = config.debug.capture_mode
        self.graph = torch.cuda.CUDAGraph()
        if self.split_graphs:
            # two graphs sharing one memory pool: the collective of the first gradient segment is issued between them
            self.graph_b = torch.cuda.CUDAGraph()
            ok = False
            try:
                with torch.cuda.graph(self.graph, capture_error_mode=mode):
                    self.losses, mf = self._stage_a(self.static_batch)
                with torch.cuda.graph(self.graph_b, pool=self.graph.pool(), capture_error_mode=mode):
                    self._stage_b(mf)
                ok = True
            finally:
                self._stage_end(failed=not ok)
        else:
            with torch.cuda.graph(self.graph, capture_error_mode=mode):
                self.losses = self._step_pipelined(self.static_batch) if pipelined else self._fwd_bwd(self.static_batch)
        # the NaN flags of the captured step live in the graph's private pool and are rewritten by every replay: keep them
        # as persistent handles (the reference asserts on NaN at every step, loss_functions.py:60,105,115)
        self.nan_flags = LF.take_nan_flags()
        LF.pyramid_cache.clear()

    def check_finite(self):
        """Deferred NaN assert of the most recent step (one host sync); works in eager and in hipGraph mode."""
        LF.check_finite(self.nan_flags)

    def grad_norms(self):
        """L2 norm of the most recent step's (all-reduced, unscaled) gradient per network, from the flat bucket
        -> {'disp'|'pose'|'mask'|'flow': 0-dim float64 device tensor}.  Diagnostic / parity-test hook."""
        out = {}
        for name, r in zip(NET_NAMES, self.opt.net_ranges):
            if r is not None:
                out[name] = self.opt.flat_g[r[0]:r[1]].double().pow(2).sum().sqrt()
        return out

    def save_checkpoint(self, save_path, epoch, is_best=False):
        """train.py:396-413: the five ``{'epoch', 'state_dict'}`` files of utils.save_checkpoint."""
        from . import utils
        # parameters are views into the 297 MB flat bucket: torch.save would write the WHOLE storage behind every view
        # (each net file 297 MB, carrying the other nets' weights) -> clone to per-tensor storages first
        def own(n):
            return {k: v.detach().clone() for k, v in n.state_dict().items()}
        sd = [({"epoch": epoch + 1, "state_dict": own(n)} if n is not None else {"epoch": epoch + 1, "state_dict": {}})
              for n in self.nets]
        utils.save_checkpoint(save_path, sd[0], sd[1], sd[2], sd[3], {"epoch": epoch + 1, "state_dict": self.opt.state_dict()},
                              is_best)

    def segments(self):
        """[(lo, hi)] of the flat gradient bucket as exchanged: one per trainable network (per_network), or
        [DispResNet6 | PoseNetB6] [MaskNet6 | Back2Future] (post / staged)"""
        n = self.opt.flat_g.numel()
        if self.pipeline == "per_network":
            if self.segment_calls:              # as the most recent (captured) step issued them: chunks of DispResNet6's segment included
                return [(lo, hi) for _, lo, hi in self.segment_calls]
            return [s for s in (self.opt.segment(i) for i in range(len(self.nets))) if s is not None]
        return [(0, self.n_dp), (self.n_dp, n)] if 0 < self.n_dp < n else [(0, n)]

    def calibrate_comm(self, reps=3):
        """Each gradient segment's all-reduce ALONE on an idle device (median of `reps`, host-synchronised): the yardstick
        the exposed communication is read against.  Outside any timed region; needs an initialised process group."""
        if not self.opt.comm_active():
            return None
        out = []
        for lo, hi in self.segments():
            ts = []
            for _ in range(reps):
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                if self.pipeline == "per_network":
                    self.opt.all_reduce_here(lo, hi)
                else:
                    w = dist.all_reduce(self.opt.flat_g[lo:hi], async_op=True)
                    w.wait()
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            out.append(sorted(ts)[len(ts) // 2])
        self.comm_standalone_ms = out
        return out

    def stage_b_ms(self):
        """median stream time of backward stage B over the recent steps (after a device synchronise), or None"""
        if not self.stage_b_events:
            return None
        v = sorted(a.elapsed_time(b) for a, b in self.stage_b_events)
        return round(v[len(v) // 2], 3)

    def comm_stats(self):
        """What the data-parallel step exchanges and how (call after a device synchronise).  per_network: the segments in issue
        order and each one's all-reduce alone (calibrate_comm); the EXPOSED time of the exchange inside a replayed graph cannot
        be bracketed by events -- bench.py measures it as step time with the collectives minus step time with them skipped.
        post / staged: median stream time the compute stream spent waiting for each segment's all-reduce; `overlapped` = the
        standalone duration minus that."""
        if self.pipeline == "per_network":
            if not self.opt.comm_active():
                return None
            order = [NET_NAMES[i] for i, _, _ in self.segment_calls]
            r = {"design": ("per-network gradient pipelines inside ONE graph: at the end of a network's backward pass its own stream "
                            "issues ncclAllReduce on its segment of the flat bucket (a node of that graph branch), then its Adam segment, "
                            "then its weight images; issue order = order the backward passes are enqueued (shortest first); DispResNet6 "
                            "(the last finisher; every issuing stream has a communicator of its own) " +
                            ("hands its segment over in chunks (decoder, conv5-7, rest) while its backward pass still runs, their tails "
                             "on the step's origin stream" if self._chunk_lo else
                             "exchanges its segment at the end of its backward pass: the one exchange no other network's backward covers "
                             "(config.grad_chunks would start it earlier; off: it costs the one-GPU graph +1.5 ms)")),
                 "issue_order": order, "segments_mb": [round(4e-6 * (hi - lo), 1) for _, lo, hi in self.segment_calls],
                 "collective": "ncclAllReduce on the issuing stream (cc_amd/rccl.py), one communicator per issuing stream (%d)" % len(self.opt._rccl) if self.opt._rccl is not None
                 else "torch.distributed.all_reduce (blocking)"}
            if self.comm_standalone_ms and len(self.comm_standalone_ms) == len(order):
                r["standalone_ms"] = [round(v, 3) for v in self.comm_standalone_ms]
            return r
        if not self.comm_events:
            return None
        segs = self.segments()
        per = []
        for ev in self.comm_events:
            per.append([ev[2 * i].elapsed_time(ev[2 * i + 1]) for i in range(len(ev) // 2)])
        med = [sorted(col)[len(col) // 2] for col in zip(*per)]
        r = {"segments_mb": [round(4e-6 * (hi - lo), 1) for lo, hi in segs], "exposed_ms": [round(v, 3) for v in med],
             "exposed_ms_total": round(sum(med), 3), "steps_sampled": len(per),
             "design": ("two all-reduces per step (the [DispResNet6|PoseNetB6] segment is issued after backward stage A and runs "
                        "under stage B; the [MaskNet6|Back2Future] segment after stage B) in place of north_star's single "
                        "all-reduce; one all-reduce when only one segment is trainable") if self.graph_b is not None or not self.use_graph
             else ("per-network streams: ONE graph (the four networks' backward passes side by side), both segment all-reduces "
                   "issued behind it; the second one runs under the first segment's optimizer update")}
        if self.comm_standalone_ms and len(self.comm_standalone_ms) == len(med):
            r["standalone_ms"] = [round(v, 3) for v in self.comm_standalone_ms]
            r["overlapped_ms"] = round(sum(max(0.0, a - b) for a, b in zip(self.comm_standalone_ms, med)), 3)
        return r

    def step(self, batch):
        """train.py:445-568 for one mini-batch: returns the (device) loss tensors of this step."""
        if self.pipeline == "per_network":
            if self._weights_touched():
                ops.packs.mark_stale()
                if self.graph is not None:
                    ops.packs.prepack_all()     # (the captured step contains no start-of-step refresh)
                    ops.packs.end_step()
            if self.use_graph:
                if self.graph is None:
                    try:
                        self.capture(batch)
                    except RuntimeError as e:
                        if not self.opt.comm_active():
                            raise
                        # a collective that cannot be captured on this stack: the legacy form issues them outside the graph
                        self._fall_back("capturing the step with its collectives failed (%s)" % (str(e).splitlines()[0][:200],))
                        return self._step_legacy(batch)
                self._copy_in(batch)
                self.graph.replay()
                return self.losses
            return self._step_pipelined(batch)
        return self._step_legacy(batch)

    def _step_legacy(self, batch):
        opt = self.opt
        works = []

        def reduce_dp():            # 227 MB segment: in flight while MaskNet6 + Back2Future run their backward pass
            works.append(opt.all_reduce(0, self.n_dp, async_op=True))

        if self.use_graph:
            if self.graph is None:
                self.capture(batch)
            self._copy_in(batch)
            self.graph.replay()
            if self.graph_b is not None:
                reduce_dp()
                if self.comm_debug.get("events"):
                    # stage B (backward of MaskNet6 + Back2Future) runs with the big segment's all-reduce in flight: RCCL's copy / reduce
                    # workgroups compete with the MFMA kernels for CUs -- its stream time per step, to be read against the same stage
                    # with the collectives skipped (bench.py: comm.stage_b_ms)
                    eb = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                    eb[0].record()
                    self.graph_b.replay()
                    eb[1].record()
                    self.stage_b_events = (self.stage_b_events + [eb])[-64:]
                else:
                    self.graph_b.replay()
            losses = self.losses
        else:
            losses = self._fwd_bwd(batch, between=reduce_dp if (opt.comm_active() and self.split_graphs) else None)
        n = opt.flat_p.numel()
        if opt.comm_active():
            if not works:
                works.append(opt.all_reduce(0, self.n_dp, async_op=True))
            works.append(opt.all_reduce(self.n_dp, None, async_op=True))      # mask + flow segment (70 MB): exposed
            cut = self.n_dp                     # (a network boundary: 256-byte aligned)
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)] \
                if (losses["loss"].is_cuda and self.comm_debug.get("events")) else None
            join = self.comm_debug.get("join", "segmented")          # A/B (tools/gpu_r3z.sh): "single" = one join, one Adam launch
            if join == "single" and all(w is not None for w in works):
                if ev:
                    ev[0].record()
                works[-1].wait()              # the group's collectives complete in issue order on its stream: the last one covers all
                if ev:
                    ev[1].record()
                    self.comm_events = (self.comm_events + [ev[:2]])[-64:]
                opt.step(opt.grad_scale())
                return losses
            if 0 < cut < n and len(works) == 2 and all(w is not None for w in works):
                # the big segment's update runs while the small segment is still being exchanged.  work.wait() makes the
                # compute stream wait for the collective: the stream time between the events around it is the EXPOSED part
                # of that collective (what backward stage B / the first Adam segment did not hide)
                if ev:
                    ev[0].record()
                works[0].wait()
                if ev:
                    ev[1].record()
                opt.step_segment(0, cut, True, opt.grad_scale())
                if ev:
                    ev[2].record()
                works[1].wait()
                if ev:
                    ev[3].record()
                    self.comm_events = (self.comm_events + [ev])[-64:]
                opt.step_segment(cut, None, False, opt.grad_scale())
                return losses
            if ev:
                ev[0].record()
            for w in works:
                if w is not None:
                    w.wait()
            if ev:
                ev[1].record()
                self.comm_events = (self.comm_events + [ev[:2]])[-64:]
        opt.step(opt.grad_scale())                                          # :568
        return losses
