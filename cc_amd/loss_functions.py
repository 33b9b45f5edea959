"""Drop-in for the reference's ``loss_functions`` module on the fused gfx950 kernels
(cc_amd/csrc/{warp,ssim,losses}.hip, C ABI in include/ccengine.h).

Same names / signatures / return types as loss_functions.py (the 14 names train.py:23-26 imports).
MI355X-first differences in HOW, not in WHAT:

* every loss is one ``torch.autograd.Function`` whose forward launches the fused kernels AND their
  adjoints (a scalar loss with constant weights has a known upstream gradient), so nothing but the
  final input-gradients is kept between forward and backward; ``backward`` only scales them;
* all reductions are two-stage and deterministic (no float atomics), the OOB normaliser
  ``nelement/sum(valid)`` (loss_functions.py:48,103) stays on the device -> zero host syncs per step
  (the reference does ~66); the NaN asserts (:60,105,115) become a device flag, checked on demand
  (``check_finite()``) or eagerly with ``config.strict_nan_checks = True``;
* the image pyramid (``adaptive_avg_pool2d``, recomputed 90x per step by the reference) is cached
  per source tensor.

Quirks Q1-Q8 of SURVEY.md section 8 are reproduced on purpose.
"""
import torch
from torch import nn

from . import config
from ._lib import engine, STREAM
from .inverse_warp import projection_matrix, pose2flow, _ac
from .ssim import gauss13_ptr

epsilon = 1e-8


# ----------------------------------------------------------------------------- helpers
def _f32c(t):
    return t.contiguous().float()


class _ScalarPool:
    """Zero-initialised device scalars (loss accumulators, NaN flags) handed out from ONE zero-filled buffer per training
    step instead of one fill launch each (~25 per step).  Opt-in: only a trainer that brackets its forward+backward with
    begin()/end() gets pooled scalars (under hipGraph capture the fill is then part of the graph); everybody else gets a
    fresh torch.zeros(1)."""
    SIZE = 128

    def __init__(self):
        self.buf, self.used = None, 0

    def begin(self, device):
        self.buf = torch.zeros(self.SIZE, device=device, dtype=torch.float32)
        self.used = 0

    def end(self):
        self.buf = None

    def take(self, ref):
        if self.buf is None or self.used >= self.SIZE or self.buf.device != ref.device:
            return torch.zeros(1, device=ref.device, dtype=torch.float32)
        v = self.buf[self.used:self.used + 1]
        self.used += 1
        return v


scalar_pool = _ScalarPool()


def _zeros1(ref):
    return scalar_pool.take(ref)


def _empty(n, ref):
    return torch.empty(max(int(n), 1), device=ref.device, dtype=torch.float32)


class _Jobs:
    """Host-side job table of a *_jobs entry point: rows of 8 slots (tensors -> device addresses, ints, None) + (H, W)."""

    def __init__(self):
        self.rows = []

    def add(self, slots, H, W):
        self.rows.append((list(slots) + [None] * (8 - len(slots)), int(H), int(W)))

    def __len__(self):
        return len(self.rows)

    def pack(self):
        import ctypes
        E = engine()
        arr = (ctypes.c_long * (10 * len(self.rows)))()
        for j, (slots, H, W) in enumerate(self.rows):
            for k, v in enumerate(slots):
                if v is None:
                    arr[10 * j + k] = 0
                elif torch.is_tensor(v):
                    arr[10 * j + k] = E._ptr(v, "job", k)
                else:
                    arr[10 * j + k] = int(v)
            arr[10 * j + 8], arr[10 * j + 9] = H, W
        self._keep = arr
        return ctypes.addressof(arr)


def _off(t, nfloats):
    """Device address of element `nfloats` of a contiguous fp32 tensor."""
    return engine()._ptr(t, "job", 0) + 4 * int(nfloats)


MAX_JOBS = 24

_nan_flags = []
_sticky_nan = {}          # device -> 1-element flag: OR of the flags that left the bounded list unchecked


def check_finite(persistent=()):
    """Deferred form of the reference's ``assert((loss == loss).item() == 1)`` (one host sync).  `persistent`: flag
    tensors that a captured hipGraph step rewrites at every replay (CCTrainer keeps them; they are never dropped)."""
    flags = list(_nan_flags) + list(persistent) + list(_sticky_nan.values())
    del _nan_flags[:]
    _sticky_nan.clear()
    bad = bool(torch.stack([f.reshape(()) for f in flags]).ne(0).any().item()) if flags else False
    assert not bad, "NaN encountered in a photometric loss term"


def take_nan_flags():
    """Hand the flags registered since the last check to the caller (the trainer, right after capturing a step)."""
    out = list(_nan_flags)
    del _nan_flags[:]
    return out


def _register_nan_flag(flag):
    if config.strict_nan_checks:
        assert flag.item() == 0, "NaN encountered in a photometric loss term"
    else:
        _nan_flags.append(flag)
        if len(_nan_flags) > 96 and not (flag.is_cuda and torch.cuda.is_current_stream_capturing()):
            # a caller that checks rarely (or never): the oldest flags are OR-ed into one sticky flag per device instead of
            # being dropped, so that an early NaN still fails the next check_finite()
            old, keep = _nan_flags[:-64], _nan_flags[-64:]
            del _nan_flags[:]
            _nan_flags.extend(keep)
            for dev in {f.device for f in old}:
                m = torch.stack([f.reshape(()) for f in old if f.device == dev]).ne(0).any().to(torch.float32).reshape(1)
                prev = _sticky_nan.get(dev)
                _sticky_nan[dev] = m if prev is None else torch.maximum(prev, m)


class _PyramidCache:
    """adaptive_avg_pool2d(img, (h, w)) results keyed on the identity (+ version) of the source tensor."""

    def __init__(self, cap=16):
        self.cap = cap
        self.items = {}

    def clear(self):
        self.items.clear()

    def _entry(self, img):
        key = id(img)
        ent = self.items.get(key)
        if ent is None or ent[0] is not img or ent[1] != img._version:
            ent = (img, img._version, {})
            self.items[key] = ent
            while len(self.items) > self.cap:
                self.items.pop(next(iter(self.items)))
        return ent

    def prefetch(self, imgs):
        """The 2^l pyramids of several same-sized frames (the target and the reference frames of a step) in ONE launch
        (cc_pyramid_build_multi) instead of one per frame at its first use.  Frames that do not qualify are left to get()."""
        import ctypes
        todo = []
        for img in imgs:
            H, W = img.shape[2], img.shape[3]
            if H % 32 or W % 32 or img.dtype != torch.float32 or not img.is_contiguous() or img.data_ptr() % 16:
                continue
            ent = self._entry(img)
            if (H >> 1, W >> 1) not in ent[2]:
                todo.append((img, ent))
        if len(todo) < 2 or len(todo) > 8 or len({tuple(i.shape) for i, _ in todo}) != 1:
            return
        B, C, H, W = todo[0][0].shape
        sizes = [(H >> k, W >> k) for k in range(1, 6)]
        n = sum(B * C * a * b for a, b in sizes)
        packed = [torch.empty(n, device=todo[0][0].device, dtype=torch.float32) for _ in todo]
        src = (ctypes.c_long * len(todo))(*[i.detach().data_ptr() for i, _ in todo])
        dst = (ctypes.c_long * len(todo))(*[p.data_ptr() for p in packed])
        engine().call("cc_pyramid_build_multi", ctypes.addressof(src), ctypes.addressof(dst), len(todo), 6, B * C, H, W, STREAM)
        for (img, ent), pk in zip(todo, packed):
            off = 0
            for a, b in sizes:
                ent[2][(a, b)] = pk[off:off + B * C * a * b].view(B, C, a, b)
                off += B * C * a * b

    def edge_weights(self, img, sizes):
        """{(h, w): [B,2,h,w] edge weights of the pooled image} for the given level sizes -- computed once per (image, scale) by ONE
        cc_edge_weights_jobs launch for all missing levels and shared by every smoothness term of the step."""
        ent = self._entry(img)
        miss = [hw for hw in dict.fromkeys(sizes) if ("ew",) + hw not in ent[2]]
        if miss:
            B = img.shape[0]
            jb = _Jobs()
            for (h, w) in miss:
                lv = self.get(img, h, w)
                out = torch.empty(B, 2, h, w, device=lv.device, dtype=torch.float32)
                ent[2][("ew", h, w)] = out
                jb.add([lv, out], h, w)
            engine().call("cc_edge_weights_jobs", jb.pack(), len(jb), B, STREAM)
        return {hw: ent[2][("ew",) + hw] for hw in sizes}

    def get(self, img, h, w):
        H, W = img.shape[2], img.shape[3]
        if (h, w) == (H, W):
            return _f32c(img)
        key = id(img)
        ent = self.items.get(key)
        if ent is None or ent[0] is not img or ent[1] != img._version:
            ent = (img, img._version, {})
            self.items[key] = ent
            while len(self.items) > self.cap:
                self.items.pop(next(iter(self.items)))
        lv = ent[2].get((h, w))
        if lv is None:
            src = _f32c(img.detach())
            B, C = src.shape[0], src.shape[1]
            lvl = next((k for k in range(1, 6) if (H >> k, W >> k) == (h, w)), None)
            if lvl is not None and H % 32 == 0 and W % 32 == 0:
                # the whole 2^l pyramid (levels 1..5) of this image in ONE launch: every loss scale will ask for it
                sizes = [(H >> k, W >> k) for k in range(1, 6)]
                packed = torch.empty(sum(B * C * a * b for a, b in sizes), device=src.device, dtype=torch.float32)
                engine().call("cc_pyramid_build", src, packed, 6, B * C, H, W, STREAM)
                off = 0
                for a, b in sizes:
                    ent[2][(a, b)] = packed[off:off + B * C * a * b].view(B, C, a, b)
                    off += B * C * a * b
                return ent[2][(h, w)]
            lv = torch.empty(B, C, h, w, device=src.device, dtype=torch.float32)
            engine().call("cc_adaptive_avg_pool", src, lv, B * C, H, W, h, w, STREAM)
            ent[2][(h, w)] = lv
        return lv


pyramid_cache = _PyramidCache()


def _scaled_intrinsics(intrinsics, intrinsics_inv, downscale):
    """loss_functions.py:91-92."""
    K_s = torch.cat((intrinsics[:, 0:2] / downscale, intrinsics[:, 2:]), dim=1)
    Kinv_s = torch.cat((intrinsics_inv[:, :, 0:2] * downscale, intrinsics_inv[:, :, 2:]), dim=2)
    return K_s, Kinv_s


class _HeadGrads:
    """Per-step gradient accumulators of the network outputs the loss terms share (train.py:509,567: a disparity / flow / mask
    level feeds two to four loss terms and the autograd engine adds their gradients pairwise -- ~44 launches per step).  While
    the trainer has this active (CCTrainer._stage_a: ONE backward pass per forward pass), every fused loss scales its stashed
    gradients straight INTO the tensor's accumulator (cc_scale_acc_jobs: first writer =, later writers +=, in the engine's own
    execution order); the term that registered the tensor first hands the accumulator to autograd, the others return None.  The
    engine runs a tensor's consumer only after every term that has an edge to it has executed, so the sum is complete by then.
    Inactive (stand-alone use of the loss functions, double backward): every term returns its own scaled copy, as before."""

    def __init__(self):
        self.enabled = True          # (bench.py A/B: CC_NO_HEAD_ACC=1 clears it)
        self.active = False
        self.acc = {}

    def begin(self):
        self.active, self.acc = self.enabled, {}

    def end(self):
        self.active, self.acc = False, {}

    def register(self, t, owner):
        # keyed on the autograd tensor itself (kept alive until end(), so its id is not reused): two distinct tensors that alias
        # one storage get one accumulator each and autograd adds them -- correct whichever producers they feed
        key = id(t)
        ent = self.acc.get(key)
        if ent is None:
            ent = self.acc[key] = [torch.empty(t.numel(), device=t.device, dtype=torch.float32), owner, False, t]
        return ent


head_grads = _HeadGrads()

# The stream the fused loss terms below issue their FORWARD launches on (None: the current one).  Set by cc_amd.trainer.cc_forward
# around the calls it wants beside -- not behind -- the rigid photometric loss (warps, SSIM, adjoints: a fused term computes its
# gradients in the forward call).  The switch sits INSIDE forward on purpose: the call returns on the caller's stream, so autograd binds
# the node -- and with it the backward call, which adds into the step's shared gradient accumulators (_HeadGrads) in the engine's
# execution order -- to the caller's stream; a `with torch.cuda.stream(..)` AROUND the call would move the backward call to the side
# stream and race with the other terms' accumulations.  The caller forks the stream before and joins it after.
forward_stream = None


def _on_forward_stream(fwd):
    def forward(ctx, *args):
        st = forward_stream
        if st is None:
            return fwd(ctx, *args)
        cur = torch.cuda.current_stream()
        with torch.cuda.stream(st):
            out = fwd(ctx, *args)
        if not torch.cuda.is_current_stream_capturing() and getattr(ctx, "arena", None) is not None:
            ctx.arena.flat.record_stream(cur)        # (eager mode: the stash is read by backward on the caller's stream)
        return out
    return forward


class _GradArena:
    """The gradients a fused loss stashes in forward live in ONE flat buffer (one view per differentiable input, same
    shape), so that backward is ONE `* grad_output` launch for the whole loss instead of one per tensor."""

    def __init__(self, inputs, need):
        ref = next(t for t in inputs if torch.is_tensor(t))
        sizes = [t.numel() if (torch.is_tensor(t) and nd) else 0 for t, nd in zip(inputs, need)]
        self.flat = torch.empty(max(sum(sizes), 1), device=ref.device, dtype=torch.float32)
        self.spans, off = [], 0
        for t, n in zip(inputs, sizes):
            self.spans.append((off, n, tuple(t.shape)) if n else None)
            off += n
        # shared per-tensor accumulators of the step (see _HeadGrads): one entry per span, or None
        self.heads = [head_grads.register(t, id(self)) if sp is not None else None
                      for t, sp in zip(inputs, self.spans)] if head_grads.active else None
        self.spent = False

    def view(self, i, flat=None):
        sp = self.spans[i]
        if sp is None:
            return None
        return (self.flat if flat is None else flat)[sp[0]:sp[0] + sp[1]].view(sp[2])

    def scaled(self, gout):
        """-> [grad_i * gout or None] (fresh storage: the arena itself stays valid for a second backward)."""
        if self.heads is not None:
            import ctypes
            if self.spent:
                raise RuntimeError("second backward through a loss term while the step's gradient accumulators are active "
                                   "(head_grads): the accumulators would be added into twice")
            self.spent = True
            # rounds[r] = the r-th occurrence of every accumulator in THIS term.  Jobs of one launch are unordered, so a tensor that
            # sits at two positions of a term (smooth_loss([m, m])) must not have its '=' and its '+=' in the same launch.
            rounds, res, seen = [[]], [], {}
            for sp, ent in zip(self.spans, self.heads):
                if sp is None:
                    res.append(None)
                    continue
                r = seen.get(id(ent), 0)
                seen[id(ent)] = r + 1
                while len(rounds) <= r:
                    rounds.append([])
                rounds[r] += [self.flat.data_ptr() + 4 * sp[0], ent[0].data_ptr(), sp[1], 1 if ent[2] else 0]
                ent[2] = True
                # the accumulator goes to autograd once: from the term that registered the tensor first, at its first position
                res.append(ent[0].view(sp[2]) if (ent[1] == id(self) and r == 0) else None)
            g1 = _f32c(gout).reshape(1)
            for jobs in rounds:
                if jobs:
                    arr = (ctypes.c_long * len(jobs))(*jobs)
                    engine().call("cc_scale_acc_jobs", ctypes.addressof(arr), len(jobs) // 4, g1, STREAM)
            return res
        out = torch.empty_like(self.flat)
        engine().call("cc_scale_by_scalar", self.flat, _f32c(gout).reshape(1), out, self.flat.numel(), STREAM)
        return [self.view(i, out) for i in range(len(self.spans))]


# ----------------------------------------------------------------------------- engine extensions: per-scale glue of train.py
# (not part of the reference's module surface; cc_amd.trainer.cc_forward uses them in place of the per-scale Python list
# comprehensions of train.py:458,470-471,475-476,488 -- same values, one launch for all scales)
def _ew_jobs(op, rows, planes, c0=0, nc=0, MC=0):
    jb = _Jobs()
    for slots, h, w in rows:
        jb.add(slots, h, w)
    engine().call("cc_elementwise_jobs", jb.pack(), len(jb), planes, op, c0, nc, MC, STREAM)


class _RecipLevelsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, *xs):
        xs = [_f32c(x) for x in xs]
        ys = [torch.empty_like(x) for x in xs]
        _ew_jobs(0, [([x, y], x.shape[2], x.shape[3]) for x, y in zip(xs, ys)], xs[0].shape[0] * xs[0].shape[1])
        ctx.save_for_backward(*ys)
        ctx.set_materialize_grads(False)
        return tuple(ys)

    @staticmethod
    def backward(ctx, *gs):
        ys = ctx.saved_tensors
        out = [None] * len(ys)
        rows = []
        for k, (g, y) in enumerate(zip(gs, ys)):
            if g is not None:
                out[k] = torch.empty_like(y)
                rows.append(([_f32c(g), y, out[k]], y.shape[2], y.shape[3]))
        if rows:
            _ew_jobs(1, rows, ys[0].shape[0] * ys[0].shape[1])
        return tuple(out)


def reciprocal_levels(xs):
    """[1 / x for x in xs] (train.py:458 depth = 1 / disparity) for same-(B, C) maps of all scales in one launch."""
    xs = list(xs)
    if len(xs) > MAX_JOBS or len({(x.shape[0], x.shape[1]) for x in xs}) != 1:
        return [1 / x for x in xs]
    return list(_RecipLevelsFn.apply(*xs))


def abs_diff_levels(a_list, b_list):
    """[(a - b).abs() ...] WITHOUT gradient (train.py:475-476: the rigidity masks only ever enter `< THRESH` comparisons)."""
    with torch.no_grad():
        a_list, b_list = [_f32c(a) for a in a_list], [_f32c(b) for b in b_list]
        if len(a_list) > MAX_JOBS or len({(a.shape[0], a.shape[1]) for a in a_list}) != 1:
            return [(a - b).abs() for a, b in zip(a_list, b_list)]
        outs = [torch.empty_like(a) for a in a_list]
        _ew_jobs(2, [([a, b, o], a.shape[2], a.shape[3]) for a, b, o in zip(a_list, b_list, outs)],
                 a_list[0].shape[0] * a_list[0].shape[1])
        return outs


def imagenet_normalize_levels(ims):
    """[((im * 0.5 + 0.5) - mean) / std for im in ims] (Back2Future.normalize, back2future.py:118-132) for 3-channel images that
    need no gradient, in one launch."""
    with torch.no_grad():
        ims = [_f32c(im) for im in ims]
        outs = [torch.empty_like(im) for im in ims]
        _ew_jobs(5, [([im, o], im.shape[2], im.shape[3]) for im, o in zip(ims, outs)], ims[0].shape[0] * 3)
        return outs


class _ComplementSliceFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, c0, c1, *ms):
        ms = [_f32c(m) for m in ms]
        B, MC, nc = ms[0].shape[0], ms[0].shape[1], c1 - c0
        outs = [torch.empty(B, nc, m.shape[2], m.shape[3], device=m.device, dtype=torch.float32) for m in ms]
        _ew_jobs(3, [([m, o], m.shape[2], m.shape[3]) for m, o in zip(ms, outs)], B * nc, c0, nc, MC)
        ctx.geom = (c0, nc, MC, B, [tuple(m.shape) for m in ms])
        ctx.set_materialize_grads(False)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gs):
        c0, nc, MC, B, shapes = ctx.geom
        out = [None] * len(shapes)
        rows = []
        for k, g in enumerate(gs):
            if g is not None:
                out[k] = torch.empty(shapes[k], device=g.device, dtype=torch.float32)
                rows.append(([_f32c(g), out[k]], shapes[k][2], shapes[k][3]))
        if rows:
            _ew_jobs(4, rows, B * MC, c0, nc, MC)
        return (None, None) + tuple(out)


def complement_slice_levels(masks, c0, c1):
    """[1 - m[:, c0:c1] for m in masks] (train.py:488 flow_exp_mask) for all scales in one launch (and one for the backward)."""
    masks = list(masks)
    if len(masks) > MAX_JOBS or len({(m.shape[0], m.shape[1]) for m in masks}) != 1:
        return [1 - m[:, c0:c1] for m in masks]
    return list(_ComplementSliceFn.apply(int(c0), int(c1), *masks))


def rigid_flows_levels(depths, pose, ref_ids, intrinsics, intrinsics_inv):
    """[[pose2flow(d.squeeze(1), pose[:, r], K, Kinv) for d in depths] for r in ref_ids] WITHOUT gradient (train.py:470-471:
    flows_cam_fwd / flows_cam_bwd feed thresholds and the non-differentiable consensus target only) -- one pose launch + one
    launch for all scales and reference frames."""
    import ctypes
    E = engine()
    with torch.no_grad():
        S, R, B = len(depths), pose.shape[1], pose.shape[0]
        if S * len(ref_ids) > MAX_JOBS:
            return [[pose2flow(d.squeeze(1), pose[:, r], intrinsics, intrinsics_inv) for d in depths] for r in ref_ids]
        one = (ctypes.c_float * 1)(1.0)
        P = torch.empty(1, R, B, 12, device=pose.device, dtype=torch.float32)
        Kinv_c = _f32c(intrinsics_inv)
        E.call("cc_pose_proj_levels", _f32c(pose), _f32c(intrinsics), P, 1, R, B, ctypes.addressof(one), STREAM)
        out, jb = [], _Jobs()
        for r in ref_ids:
            fl = []
            for dpt in depths:
                dd = _f32c(dpt.detach()[:, 0])
                f = torch.empty(B, 2, dd.shape[1], dd.shape[2], device=pose.device, dtype=torch.float32)
                jb.add([dd, P[0][r], Kinv_c, f], dd.shape[1], dd.shape[2])
                fl.append(f)
            out.append(fl)
        E.call("cc_pose2flow_fwd_jobs", jb.pack(), len(jb), B, 0, STREAM)
        return out


# ----------------------------------------------------------------------------- small public helpers
def spatial_normalize(disp):
    """loss_functions.py:13-16."""
    _mean = disp.mean(dim=1, keepdim=True).mean(dim=2, keepdim=True).mean(dim=3, keepdim=True)
    return disp / _mean


def robust_l1(x, q=0.5, eps=1e-2):
    """loss_functions.py:18-21."""
    return torch.pow((x.pow(2) + eps), q).mean()


def robust_l1_per_pix(x, q=0.5, eps=1e-2):
    """loss_functions.py:23-25."""
    return torch.pow((x.pow(2) + eps), q)


def logical_or(a, b):
    """loss_functions.py:157-158."""
    return 1 - (1 - a) * (1 - b)


def occlusion_masks(flow_bw, flow_fw):
    """loss_functions.py:343-352 -> (occ_bw, occ_fw), each [B,H,W] in {0,1}."""
    flow_bw, flow_fw = _f32c(flow_bw.detach()), _f32c(flow_fw.detach())
    B, _, H, W = flow_bw.shape
    no = torch.empty(B, 1, H, W, device=flow_bw.device, dtype=torch.float32)
    engine().call("cc_flow_noocc", flow_bw, flow_fw, no, B, H, W, STREAM)
    occ = (1 - no)[:, 0]
    return occ, occ.clone()


def _rigid_noocc(depth_bhw, P_full, Kinv):
    """(1 - depth_occlusion_masks) [B,4,h,w].  P_full: the four [B,3,4] projection matrices, or already stacked [4,B,3,4]."""
    B, h, w = depth_bhw.shape
    P4 = P_full if torch.is_tensor(P_full) else torch.stack(list(P_full)).contiguous()
    no = torch.empty(B, 4, h, w, device=depth_bhw.device, dtype=torch.float32)
    engine().call("cc_rigid_noocc_fused", depth_bhw, P4, Kinv, no, B, h, w, STREAM)
    return no


def depth_occlusion_masks(depth, pose, intrinsics, intrinsics_inv):
    """loss_functions.py:132-137 -> [B,4,h,w] occlusion masks (needs exactly 4 reference poses, H3)."""
    d = _f32c(depth.detach().squeeze())
    assert d.dim() == 3, "wrong size for depth"           # loss_functions.py:133 squeezes the batch away for B=1 (H4)
    Kinv = _f32c(intrinsics_inv.detach())
    P_full = [_f32c(projection_matrix(pose[:, i].detach(), intrinsics.detach())) for i in range(pose.size(1))]
    return 1 - _rigid_noocc(d, P_full, Kinv)


# ----------------------------------------------------------------------------- photometric losses
class _PhotoCfg:
    def __init__(self, **kw):
        self.__dict__.update(kw)


def _photo_term(E, tgt_s, warped, mask_a, a_bs, mask_b, b_bs, gmask, gm_bs, want_grad, cfg, loss_acc, nan_flag, scale=None):
    """One (scale, reference) term: fused forward (+ adjoint maps).  Returns the scratch needed by the adjoint."""
    B, _, h, w = tgt_s.shape
    nblk = E.call("cc_ssim_num_blocks", B, h, w)
    partials = _empty(nblk * 4, tgt_s)
    if scale is None:
        scale = _empty(1, tgt_s)
    if want_grad:
        adj = [torch.empty_like(warped) for _ in range(4)]
    else:
        adj = [None] * 4
    E.call("cc_ssim_photo_fwd", tgt_s, warped, mask_a, a_bs, mask_b, b_bs, 0, partials, adj[0], adj[1], adj[2], adj[3],
           gmask, gm_bs, 1 if want_grad else 0, float(cfg.wssim), float(cfg.qch), float(cfg.lambda_oob), loss_acc,
           scale, nan_flag, gauss13_ptr(), B, h, w, STREAM)
    return adj, scale


_colscale_cache = {}


def _kinv_levels(Kinv_c, downs):
    """intrinsics_inv[:, :, 0:2] * downscale for every level (loss_functions.py:92) in one launch -> [S,B,3,3]."""
    key = (str(Kinv_c.device), tuple(downs))
    cs = _colscale_cache.get(key)
    if cs is None:
        cs = torch.tensor([[d, d, 1.0] for d in downs], dtype=torch.float32).to(Kinv_c.device).view(len(downs), 1, 1, 3)
        _colscale_cache[key] = cs
    return (Kinv_c.unsqueeze(0) * cs).contiguous()


def _carve(sizes, ref):
    """One allocation, consecutive views of the given element counts."""
    flat = torch.empty(max(sum(sizes), 1), device=ref.device, dtype=torch.float32)
    out, off = [], 0
    for n in sizes:
        out.append(flat[off:off + n])
        off += n
    return out


def _photo_rigid_jobs(ctx, cfg, tgt_img, intrinsics, intrinsics_inv, pose, refs, depths, masks, need):
    """loss_functions.py:80-128 with every (scale, reference frame) term of a pass in ONE launch (job tables, csrc/jobs.h):
    ~10 launches for the 24 terms (pose -> P, occlusion masks, warp, SSIM + robust-L1 + masks, finalize, SSIM adjoint, warp
    backward, pose gradient, per-scale sums) instead of ~14 per term."""
    import ctypes
    E = engine()
    R, S = cfg.n_refs, cfg.n_scales
    B = tgt_img.shape[0]
    want_grad = any(need)
    loss_acc, nan_flag = _zeros1(tgt_img), _zeros1(tgt_img)
    arena = _GradArena([pose] + list(refs) + list(depths) + list(masks), [need[4]] + [False] * R + list(need[5 + R:]))
    hw = []
    for s in range(S):
        assert masks[s] is None or depths[s].size()[2:] == masks[s].size()[2:]
        assert pose.size(1) == R
        hw.append((depths[s].shape[2], depths[s].shape[3]))
    px = [h * w for h, w in hw]
    downs = [tgt_img.size(2) / h for h, _ in hw]
    pose_c, K_c, Kinv_c = _f32c(pose.detach()), _f32c(intrinsics.detach()), _f32c(intrinsics_inv.detach())
    kdiv = (ctypes.c_float * (S + 1))(*(downs + [1.0]))
    P_all = torch.empty(S + 1, R, B, 12, device=tgt_img.device, dtype=torch.float32)
    E.call("cc_pose_proj_levels", pose_c, K_c, P_all, S + 1, R, B, ctypes.addressof(kdiv), STREAM)
    Kinv_all = _kinv_levels(Kinv_c, downs)
    d = [_f32c(depths[s].detach()[:, 0]) for s in range(S)]
    m = [None if masks[s] is None else _f32c(masks[s].detach()) for s in range(S)]
    tgt_l = [pyramid_cache.get(tgt_img, h, w) for h, w in hw]
    ref_l = [[pyramid_cache.get(refs[r], h, w) for h, w in hw] for r in range(R)]
    # ---- occlusion masks of all scales (full-resolution K at every scale, Q4)
    no = _carve([B * 4 * n for n in px], tgt_img)
    jb = _Jobs()
    for s in range(S):
        jb.add([d[s], P_all[S], Kinv_c, no[s]], *hw[s])
    E.call("cc_rigid_noocc_jobs", jb.pack(), len(jb), B, STREAM)
    # ---- warps
    warped = _carve([B * 3 * px[s] for s in range(S) for _ in range(R)], tgt_img)
    jb = _Jobs()
    for s in range(S):
        for r in range(R):
            jb.add([ref_l[r][s], d[s], P_all[s][r], Kinv_all[s], warped[s * R + r]], *hw[s])
    E.call("cc_inverse_warp_fwd_jobs", jb.pack(), len(jb), B, 3, cfg.border, cfg.ac, STREAM)
    # ---- fused SSIM + robust-L1 + masks, partial sums, adjoint maps; finalize (loss, per-term normalisers, NaN flag)
    nblk = [E.call("cc_ssim_num_blocks", B, h, w) for h, w in hw]
    partials = _carve([4 * nblk[s] for s in range(S) for _ in range(R)], tgt_img)
    adj = _carve([4 * B * 3 * px[s] for s in range(S) for _ in range(R)], tgt_img) if want_grad else None
    gm = [arena.view(1 + R + S + s) if (m[s] is not None and want_grad) else None for s in range(S)]
    scales = torch.empty(S * R, device=tgt_img.device, dtype=torch.float32)
    jb = _Jobs()
    for s in range(S):
        MC = 0 if m[s] is None else m[s].shape[1]
        for r in range(R):
            jb.add([tgt_l[s], warped[s * R + r], _off(no[s], r * px[s]),
                    None if m[s] is None else _off(m[s], r * px[s]),
                    None if gm[s] is None else _off(gm[s], r * px[s]),
                    None if adj is None else adj[s * R + r], partials[s * R + r], 4 | (MC << 8) | (MC << 16)], *hw[s])
    E.call("cc_ssim_photo_fwd_jobs", jb.pack(), len(jb), B, 0, 1 if want_grad else 0, float(cfg.wssim), float(cfg.qch),
           float(cfg.lambda_oob), loss_acc, scales, nan_flag, gauss13_ptr(), STREAM)
    if want_grad:
        gw = _carve([B * 3 * px[s] for s in range(S) for _ in range(R)], tgt_img)
        jb = _Jobs()
        for s in range(S):
            for r in range(R):
                j = s * R + r
                jb.add([adj[j], tgt_l[s], warped[j], _off(scales, j), gw[j]], *hw[s])
        E.call("cc_ssim_photo_bwd_jobs", jb.pack(), len(jb), B, gauss13_ptr(), STREAM)
        gd_all = _carve([R * B * px[s] for s in range(S)], tgt_img)
        nb = [(n + 255) // 256 for n in px]
        gpp = _carve([B * nb[s] * 12 for s in range(S) for _ in range(R)], tgt_img)
        jb, jp = _Jobs(), _Jobs()
        for s in range(S):
            for r in range(R):
                j = s * R + r
                jb.add([gw[j], ref_l[r][s], d[s], P_all[s][r], Kinv_all[s], _off(gd_all[s], r * B * px[s]), gpp[j]], *hw[s])
                jp.add([gpp[j]], *hw[s])
        E.call("cc_inverse_warp_bwd_jobs", jb.pack(), len(jb), B, 3, cfg.border, cfg.ac, STREAM)
        if need[4]:
            E.call("cc_pose_grad_jobs", jp.pack(), len(jp), S, R, B, pose_c, K_c, arena.view(0), ctypes.addressof(kdiv), STREAM)
        jb = _Jobs()
        for s in range(S):
            gdv = arena.view(1 + R + s)
            if gdv is None and gm[s] is None:
                continue
            jb.add([gd_all[s] if gdv is not None else None, gdv, gm[s], _off(scales, s * R)], *hw[s])
        if len(jb):
            E.call("cc_sum_refs_scale_jobs", jb.pack(), len(jb), B, R, R, STREAM)
    _register_nan_flag(nan_flag)
    ctx.arena = arena
    ctx.need = need
    return loss_acc.reshape(())


class _PhotoRigidFn(torch.autograd.Function):
    """loss_functions.py:80-128 over all scales and reference frames."""

    @staticmethod
    def forward(ctx, cfg, tgt_img, intrinsics, intrinsics_inv, pose, *rest):
        R, S = cfg.n_refs, cfg.n_scales
        refs = rest[:R]
        depths = rest[R:R + S]
        masks = rest[R + S:]
        E = engine()
        dev = tgt_img.device
        need = ctx.needs_input_grad
        if (cfg.rotation_mode == 'euler' and R * S <= MAX_JOBS and R == 4
                and all(mk is None or mk.shape[1] == R for mk in masks)):
            return _photo_rigid_jobs(ctx, cfg, tgt_img, intrinsics, intrinsics_inv, pose, refs, depths, masks, need)
        want_grad = any(need)
        loss_acc, nan_flag = _zeros1(tgt_img), _zeros1(tgt_img)
        Kinv_full = _f32c(intrinsics_inv.detach())
        with torch.enable_grad():
            pose_l = pose.detach().requires_grad_(bool(need[4]))
            K_d = intrinsics.detach()
            P_full = [projection_matrix(pose_l[:, r], K_d, cfg.rotation_mode) for r in range(R)]
        P_full_c = torch.stack([_f32c(p.detach()) for p in P_full]).contiguous()      # [4,B,3,4], shared by all scales
        direct_pose = cfg.rotation_mode == 'euler'      # HIP pose->P adjoint instead of a torch graph over 17 tiny ops
        # stash layout = the differentiable inputs: pose, refs (no gradient), depths, masks
        arena = _GradArena([pose] + list(rest), [need[4]] + [False] * R + list(need[5 + R:]))
        gpose_acc = None
        if want_grad and need[4] and direct_pose:
            gpose_acc = arena.view(0).zero_()
        K_c = _f32c(K_d)
        gP_all, P_all = [], []
        for s in range(S):
            d4 = depths[s]
            assert masks[s] is None or d4.size()[2:] == masks[s].size()[2:]
            assert pose.size(1) == R
            B, _, h, w = d4.shape
            d = _f32c(d4.detach()[:, 0])
            HW = h * w
            downscale = tgt_img.size(2) / h
            tgt_s = pyramid_cache.get(tgt_img, h, w)
            no = _rigid_noocc(d, P_full_c, Kinv_full)                        # [B,4,h,w]
            K_s, Kinv_s = _scaled_intrinsics(K_d, intrinsics_inv.detach(), downscale)
            Kinv_s = _f32c(Kinv_s)
            m = None if masks[s] is None else _f32c(masks[s].detach())
            gd_all = torch.empty((R,) + tuple(d.shape), device=dev, dtype=torch.float32) if (want_grad and need[5 + R + s]) else None
            gm = arena.view(1 + R + S + s) if (m is not None and want_grad) else None
            scales = _empty(R, tgt_s)
            for r in range(R):
                ref_s = pyramid_cache.get(refs[r], h, w)
                if direct_pose:
                    with torch.no_grad():
                        P = projection_matrix(pose_l[:, r], K_d, 'euler', k_div=downscale)
                else:
                    with torch.enable_grad():
                        P = projection_matrix(pose_l[:, r], K_s, cfg.rotation_mode)
                Pc = _f32c(P.detach())
                warped = torch.empty_like(ref_s)
                E.call("cc_inverse_warp_fwd", ref_s, d, Pc, Kinv_s, warped, B, 3, h, w, cfg.border, cfg.ac, STREAM)
                adj, scale = _photo_term(
                    E, tgt_s, warped, no.view(-1)[r * HW:], 4 * HW,
                    None if m is None else m.view(-1)[r * HW:], 4 * HW,
                    None if gm is None else gm.view(-1)[r * HW:], 4 * HW, want_grad, cfg, loss_acc, nan_flag,
                    scale=scales[r:r + 1])
                if want_grad:
                    gw = torch.empty_like(warped)
                    E.call("cc_ssim_photo_bwd", adj[0], adj[1], adj[2], adj[3], tgt_s, warped, scale, gw, 0, gauss13_ptr(),
                           B, h, w, STREAM)
                    gd_r = gd_all[r] if gd_all is not None else torch.empty_like(d)
                    gP = torch.empty_like(Pc)
                    ws = _empty(E.call("cc_warp_partials_bytes", B, h, w) // 4, d)
                    E.call("cc_inverse_warp_bwd", gw, ref_s, d, Pc, Kinv_s, gd_r, gP, None, ws, B, 3, h, w, cfg.border,
                           cfg.ac, STREAM)
                    if gpose_acc is not None:
                        pv = pose_l.detach()[:, r]
                        E.call("cc_pose_proj_bwd", gP, pv.data_ptr(), pv.stride(0), K_c, gpose_acc[:, r].data_ptr(),
                               gpose_acc.stride(0), B, float(downscale), 1, STREAM)
                    else:
                        gP_all.append(gP)
                        P_all.append(P)
            if gd_all is not None:       # sum over the reference frames in the reference's order (r = 0..3), one launch
                torch.sum(gd_all, dim=0, out=arena.view(1 + R + s).view(d.shape))
            if gm is not None:           # the per-reference normaliser (known only after the reduction), one launch
                gm *= scales.view(1, R, 1, 1)
        if want_grad and need[4] and gpose_acc is None:
            arena.view(0).copy_(torch.autograd.grad(P_all, pose_l, gP_all)[0])
        _register_nan_flag(nan_flag)
        ctx.arena = arena
        ctx.need = need
        return loss_acc.reshape(())

    @staticmethod
    def backward(ctx, gout):
        grads = ctx.arena.scaled(gout)
        need = ctx.need
        out = [None, None, None, None] + grads
        return tuple(g if (g is not None and need[i]) else None for i, g in enumerate(out))


def photometric_reconstruction_loss(tgt_img, ref_imgs, intrinsics, intrinsics_inv, depth, explainability_mask, pose,
                                    rotation_mode='euler', padding_mode='zeros', lambda_oob=0, qch=0.5, wssim=0.5,
                                    align_corners=None):
    """loss_functions.py:80-128."""
    if type(explainability_mask) not in [tuple, list]:
        explainability_mask = [explainability_mask]
    if type(depth) not in [list, tuple]:
        depth = [depth]
    n = min(len(depth), len(explainability_mask))            # zip() semantics of :125
    depth, explainability_mask = list(depth[:n]), list(explainability_mask[:n])
    if pose.size(1) != 4:
        raise IndexError("depth_occlusion_masks needs 4 reference frames (loss_functions.py:132-137)")
    cfg = _PhotoCfg(n_refs=len(ref_imgs), n_scales=n, rotation_mode=rotation_mode,
                    border=1 if padding_mode == 'border' else 0, ac=_ac(align_corners), lambda_oob=lambda_oob,
                    qch=qch, wssim=wssim)
    return _PhotoRigidFn.apply(cfg, tgt_img, intrinsics, intrinsics_inv, pose, *ref_imgs, *depth, *explainability_mask)


def _photo_flow_jobs(ctx, cfg, tgt_img, refs, flows, masks, need, rest):
    """loss_functions.py:27-77, every (scale, reference frame) term of a pass in one launch (see _photo_rigid_jobs)."""
    E = engine()
    R, S = cfg.n_refs, cfg.n_scales
    B = tgt_img.shape[0]
    want_grad = any(need)
    loss_acc, nan_flag = _zeros1(tgt_img), _zeros1(tgt_img)
    arena = _GradArena(list(rest), [False] * R + list(need[2 + R:]))      # refs (no gradient), flows, masks
    fl = [[_f32c(flows[i][s].detach()) for s in range(S)] for i in range(R)]
    hw = [(fl[0][s].shape[2], fl[0][s].shape[3]) for s in range(S)]
    px = [h * w for h, w in hw]
    for s in range(S):
        assert masks[s] is None or fl[0][s].size()[2:] == masks[s].size()[2:]
    m = [None if masks[s] is None else _f32c(masks[s].detach()) for s in range(S)]
    tgt_l = [pyramid_cache.get(tgt_img, h, w) for h, w in hw]
    ref_l = [[pyramid_cache.get(refs[i], h, w) for h, w in hw] for i in range(R)]
    no = _carve([B * n for n in px], tgt_img)
    jb = _Jobs()
    for s in range(S):
        jb.add([fl[0][s], fl[1][s], no[s]], *hw[s])                        # occlusion_masks(flow[0], flow[1]), :70
    E.call("cc_flow_noocc_jobs", jb.pack(), len(jb), B, STREAM)
    warped = _carve([B * 3 * px[s] for s in range(S) for _ in range(R)], tgt_img)
    jb = _Jobs()
    for s in range(S):
        for i in range(R):
            jb.add([ref_l[i][s], fl[i][s], warped[s * R + i]], *hw[s])
    E.call("cc_flow_warp_fwd_jobs", jb.pack(), len(jb), B, 3, 0, cfg.ac, STREAM)
    nblk = [E.call("cc_ssim_num_blocks", B, h, w) for h, w in hw]
    partials = _carve([4 * nblk[s] for s in range(S) for _ in range(R)], tgt_img)
    adj = _carve([4 * B * 3 * px[s] for s in range(S) for _ in range(R)], tgt_img) if want_grad else None
    gm = []
    for s in range(S):
        g = arena.view(R + R * S + s) if (m[s] is not None and want_grad) else None
        if g is None and m[s] is not None and want_grad:
            g = torch.empty_like(m[s])                 # the mask itself needs no gradient: scratch for the kernel
        gm.append(g)
    scales = torch.empty(S * R, device=tgt_img.device, dtype=torch.float32)
    jb = _Jobs()
    for s in range(S):
        MC = 0 if m[s] is None else m[s].shape[1]
        for i in range(R):
            jb.add([tgt_l[s], warped[s * R + i], no[s], None if m[s] is None else _off(m[s], i * px[s]),
                    None if gm[s] is None else _off(gm[s], i * px[s]), None if adj is None else adj[s * R + i],
                    partials[s * R + i], 1 | (MC << 8) | (MC << 16)], *hw[s])
    E.call("cc_ssim_photo_fwd_jobs", jb.pack(), len(jb), B, 0, 1 if want_grad else 0, float(cfg.wssim), float(cfg.qch),
           float(cfg.lambda_oob), loss_acc, scales, nan_flag, gauss13_ptr(), STREAM)
    if want_grad:
        gw = _carve([B * 3 * px[s] for s in range(S) for _ in range(R)], tgt_img)
        jb = _Jobs()
        for s in range(S):
            for i in range(R):
                j = s * R + i
                jb.add([adj[j], tgt_l[s], warped[j], _off(scales, j), gw[j]], *hw[s])
        E.call("cc_ssim_photo_bwd_jobs", jb.pack(), len(jb), B, gauss13_ptr(), STREAM)
        jb = _Jobs()
        for s in range(S):
            for i in range(R):
                gf = arena.view(R + i * S + s)
                if gf is not None:
                    jb.add([gw[s * R + i], ref_l[i][s], fl[i][s], gf], *hw[s])
        if len(jb):
            E.call("cc_flow_warp_bwd_jobs", jb.pack(), len(jb), B, 3, 0, cfg.ac, STREAM)
        jb = _Jobs()
        for s in range(S):
            if gm[s] is not None:
                jb.add([None, None, gm[s], _off(scales, s * R)], *hw[s])
        if len(jb):
            E.call("cc_sum_refs_scale_jobs", jb.pack(), len(jb), B, R, R, STREAM)      # per-reference normalisers
    _register_nan_flag(nan_flag)
    ctx.arena = arena
    ctx.need = need
    return loss_acc.reshape(())


class _PhotoFlowFn(torch.autograd.Function):
    """loss_functions.py:27-77 over all scales; flows = [flow list of ref 0, flow list of ref 1]."""

    @staticmethod
    @_on_forward_stream
    def forward(ctx, cfg, tgt_img, *rest):
        R, S = cfg.n_refs, cfg.n_scales
        refs = rest[:R]
        flows = [rest[R + i * S:R + (i + 1) * S] for i in range(R)]
        masks = rest[R + R * S:]
        E = engine()
        need = ctx.needs_input_grad
        if R * S <= MAX_JOBS and all(mk is None or mk.shape[1] == R for mk in masks):
            return _photo_flow_jobs(ctx, cfg, tgt_img, refs, flows, masks, need, rest)
        want_grad = any(need)
        loss_acc, nan_flag = _zeros1(tgt_img), _zeros1(tgt_img)
        arena = _GradArena(list(rest), [False] * R + list(need[2 + R:]))      # refs (no gradient), flows, masks
        for s in range(S):
            fl = [_f32c(flows[i][s].detach()) for i in range(R)]
            B, _, h, w = fl[0].shape
            HW = h * w
            assert masks[s] is None or fl[0].size()[2:] == masks[s].size()[2:]
            tgt_s = pyramid_cache.get(tgt_img, h, w)
            no = torch.empty(B, 1, h, w, device=tgt_img.device, dtype=torch.float32)
            E.call("cc_flow_noocc", fl[0], fl[1], no, B, h, w, STREAM)        # occlusion_masks(flow[0], flow[1]), :70
            m = None if masks[s] is None else _f32c(masks[s].detach())
            MC = 0 if m is None else m.shape[1]
            gm = arena.view(R + R * S + s) if (m is not None and want_grad) else None
            if gm is None and m is not None and want_grad:
                gm = torch.empty_like(m)                      # the mask itself needs no gradient: scratch for the kernel
            scales = torch.ones(max(MC, R), device=tgt_s.device, dtype=torch.float32) if MC > R else _empty(R, tgt_s)
            for i in range(R):
                ref_s = pyramid_cache.get(refs[i], h, w)
                warped = torch.empty_like(ref_s)
                E.call("cc_flow_warp_fwd", ref_s, fl[i], warped, B, 3, h, w, 0, cfg.ac, STREAM)
                adj, scale = _photo_term(
                    E, tgt_s, warped, no, HW,
                    None if m is None else m.view(-1)[i * HW:], MC * HW,
                    None if gm is None else gm.view(-1)[i * HW:], MC * HW, want_grad, cfg, loss_acc, nan_flag,
                    scale=scales[i:i + 1])
                if want_grad:
                    gw = torch.empty_like(warped)
                    E.call("cc_ssim_photo_bwd", adj[0], adj[1], adj[2], adj[3], tgt_s, warped, scale, gw, 0, gauss13_ptr(),
                           B, h, w, STREAM)
                    gf = arena.view(R + i * S + s)
                    if gf is None:
                        gf = torch.empty_like(fl[i])
                    E.call("cc_flow_warp_bwd", gw, ref_s, fl[i], gf, None, B, 3, h, w, 0, cfg.ac, STREAM)
            if gm is not None:
                if MC > R:
                    gm[:, R:] = 0
                gm *= scales[:MC].view(1, MC, 1, 1)           # per-reference normalisers, one launch
        _register_nan_flag(nan_flag)
        ctx.arena = arena
        ctx.need = need
        return loss_acc.reshape(())

    @staticmethod
    def backward(ctx, gout):
        grads = ctx.arena.scaled(gout)
        need = ctx.need
        out = [None, None] + grads
        return tuple(g if (g is not None and need[i]) else None for i, g in enumerate(out))


def photometric_flow_loss(tgt_img, ref_imgs, flows, explainability_mask, lambda_oob=0, qch=0.5, wssim=0.5,
                          align_corners=None):
    """loss_functions.py:27-77."""
    if type(flows[0]) not in [tuple, list]:
        if explainability_mask is not None:
            explainability_mask = [explainability_mask]
        flows = [[uv] for uv in flows]
    S = len(flows[0])
    if explainability_mask is None:
        explainability_mask = [None] * S
    assert len(flows) == len(ref_imgs)
    if len(flows) != 2:
        raise IndexError("occlusion_masks needs exactly two flows (loss_functions.py:70)")
    cfg = _PhotoCfg(n_refs=len(ref_imgs), n_scales=S, ac=_ac(align_corners), lambda_oob=lambda_oob, qch=qch, wssim=wssim)
    flat = [f for uv in flows for f in uv]
    return _PhotoFlowFn.apply(cfg, tgt_img, *ref_imgs, *flat, *list(explainability_mask)[:S])


# ----------------------------------------------------------------------------- mask / smoothness losses
def gaussian_explainability_loss(mask):
    """loss_functions.py:139-145 (imported by train.py:24, never called): stock torch."""
    if type(mask) not in [tuple, list]:
        mask = [mask]
    loss = 0
    for mask_scaled in mask:
        loss += torch.exp(-torch.mean((mask_scaled - 0.5).pow(2)) / 0.15)
    return loss


class _PerScaleFn(torch.autograd.Function):
    """Shared driver: sum over scales of a fused value+gradient kernel on one prediction list."""

    @staticmethod
    @_on_forward_stream
    def forward(ctx, launch, *preds):
        need = ctx.needs_input_grad
        loss_acc = _zeros1(preds[0])
        arena = _GradArena(list(preds), list(need[1:]))
        for s, p in enumerate(preds):
            launch(s, _f32c(p.detach()), arena.view(s), loss_acc)
        ctx.arena = arena
        return loss_acc.reshape(())

    @staticmethod
    def backward(ctx, gout):
        return (None,) + tuple(ctx.arena.scaled(gout))


class _ScaleJobsFn(torch.autograd.Function):
    """Sum over scales of a fused value+gradient kernel, ALL scales in one launch + one finalize (job tables).
    build(preds_detached, grad_views, partial_offsets_fn) -> issues the call; see the users below."""

    @staticmethod
    @_on_forward_stream
    def forward(ctx, issue, *preds):
        need = ctx.needs_input_grad
        loss_acc = _zeros1(preds[0])
        arena = _GradArena(list(preds), list(need[1:]))
        issue([_f32c(p.detach()) for p in preds], [arena.view(s) for s in range(len(preds))], loss_acc)
        ctx.arena = arena
        return loss_acc.reshape(())

    @staticmethod
    def backward(ctx, gout):
        return (None,) + tuple(ctx.arena.scaled(gout))


class _WeightedTotalFn(torch.autograd.Function):
    """train.py:509 ``loss = w1*loss_1 + w2*loss_2 + ...`` over 0-dim device scalars: stack + dot (two launches) and ONE scaling
    launch in the backward pass, instead of a multiply and an add per term in either direction."""

    @staticmethod
    def forward(ctx, wvec, *terms):
        ctx.save_for_backward(wvec)
        return torch.dot(torch.stack([t.reshape(()) for t in terms]), wvec)

    @staticmethod
    def backward(ctx, g):
        (wvec,) = ctx.saved_tensors
        return (None,) + tuple((g * wvec).unbind(0))


_WVEC = {}


def weighted_total(weights, terms):
    """sum_i weights[i] * terms[i] (python floats x 0-dim tensors).  The weight vector lives on the device, built once per
    (weights, device) -- never inside a stream capture: the trainer's eager warm-up steps come first."""
    key = (tuple(float(w) for w in weights), terms[0].device)
    if key not in _WVEC:
        _WVEC[key] = torch.tensor(key[0], dtype=torch.float32, device=key[1])
    return _WeightedTotalFn.apply(_WVEC[key], *terms)


def _partials_for(shapes, planes_of):
    """Per-job partial-sum areas laid out back to back: -> (flat buffer allocator, [float offsets])."""
    offs, tot = [], 0
    for shp in shapes:
        offs.append(tot)
        tot += ((shp[2] * shp[3] + 255) // 256) * planes_of(shp)
    return offs, tot


def explainability_loss(mask):
    """loss_functions.py:148-155: sum over scales of BCE(mask, ones)."""
    if type(mask) not in [tuple, list]:
        mask = [mask]
    E = engine()
    if len(mask) <= MAX_JOBS and len({(mk.shape[0], mk.shape[1]) for mk in mask}) == 1:
        def issue(ps, gs, acc):
            planes = ps[0].shape[0] * ps[0].shape[1]
            offs, tot = _partials_for([p.shape for p in ps], lambda shp: planes)
            part = _empty(tot, ps[0])
            jb = _Jobs()
            for p, g, o in zip(ps, gs, offs):
                jb.add([p, g, _off(part, o)], p.shape[2], p.shape[3])
            E.call("cc_bce_ones_fwd_bwd_jobs", jb.pack(), len(jb), planes, part, acc, 1.0, STREAM)
        return _ScaleJobsFn.apply(issue, *mask)

    def launch(s, p, g, acc):
        n = p.numel()
        E.call("cc_bce_ones_fwd_bwd", p, g, _empty(E.call("cc_elem_num_blocks", n), p), acc, 1.0, n, STREAM)
    return _PerScaleFn.apply(launch, *mask)


def edge_aware_smoothness_loss(img, pred_disp):
    """loss_functions.py:287-319."""
    E = engine()
    if len(pred_disp) <= MAX_JOBS and len({(p.shape[0], p.shape[1]) for p in pred_disp}) == 1:
        def issue(ps, gs, acc):
            B, C = ps[0].shape[0], ps[0].shape[1]
            offs, tot = _partials_for([p.shape for p in ps], lambda shp: B * C)
            part = _empty(tot, ps[0])
            jb = _Jobs()
            for p, g, o in zip(ps, gs, offs):
                jb.add([pyramid_cache.get(img, p.shape[2], p.shape[3]), p, g, _off(part, o)], p.shape[2], p.shape[3])
            E.call("cc_edge_smooth_fwd_bwd_jobs", jb.pack(), len(jb), B, C, part, acc, 1.0, STREAM)
        return _ScaleJobsFn.apply(issue, *pred_disp)

    def launch(s, p, g, acc):
        B, C, h, w = p.shape
        im = pyramid_cache.get(img, h, w)
        nb = E.call("cc_elem_num_blocks", h * w) * B * C
        E.call("cc_edge_smooth_fwd_bwd", im, p, g, _empty(nb, p), acc, 1.0, B, C, h, w, STREAM)
    return _PerScaleFn.apply(launch, *pred_disp)


def edge_aware_smoothness_sum(img, pred_lists):
    """sum(edge_aware_smoothness_loss(img, preds) for preds in pred_lists) -- train.py:497-501's four terms (depth, flow_fwd,
    flow_bwd, exp_mask) as ONE job table: one launch + one reduction instead of four of each, one gradient-scaling launch in the
    backward pass.  Engine extension (cc_amd.trainer.cc_forward); same per-term arithmetic, the partial sums of all terms are added
    in one fixed order."""
    flat = [p for preds in pred_lists for p in preds]
    if len(flat) > MAX_JOBS or len({p.shape[0] for p in flat}) != 1:
        t = None
        for preds in pred_lists:
            v = edge_aware_smoothness_loss(img, preds)
            t = v if t is None else t + v
        return t
    E = engine()

    def issue(ps, gs, acc):
        B = ps[0].shape[0]
        offs, tot = _partials_for([p.shape for p in ps], lambda shp: shp[0] * shp[1])
        part = _empty(tot, ps[0])
        jb = _Jobs()
        ew = pyramid_cache.edge_weights(img, [(p.shape[2], p.shape[3]) for p in ps]) if img.shape[1] == 3 else {}
        for p, g, o in zip(ps, gs, offs):
            hw = (p.shape[2], p.shape[3])
            jb.add([pyramid_cache.get(img, hw[0], hw[1]), p, g, _off(part, o), p.shape[1], ew.get(hw)], hw[0], hw[1])
        E.call("cc_edge_smooth_fwd_bwd_jobs", jb.pack(), len(jb), B, 0, part, acc, 1.0, STREAM)
    return _ScaleJobsFn.apply(issue, *flat)


def smooth_loss(pred_disp):
    """loss_functions.py:323-341."""
    if type(pred_disp) not in [tuple, list]:
        pred_disp = [pred_disp]
    E = engine()

    def launch(s, p, g, acc):
        B, C, h, w = p.shape
        nb = E.call("cc_elem_num_blocks", h * w) * B * C
        E.call("cc_smooth2_fwd_bwd", p, g, _empty(nb, p), acc, 1.0 / (2.3 ** s), 1.0, B * C, h, w, STREAM)
    return _PerScaleFn.apply(launch, *pred_disp)


# ----------------------------------------------------------------------------- consensus
def consensus_exp_masks(cam_flows_fwd, cam_flows_bwd, flows_fwd, flows_bwd, tgt_img, ref_img_fwd, ref_img_bwd, wssim,
                        wrig, ws=0.1, align_corners=None):
    """loss_functions.py:160-202: per-pixel {0,1} consensus target (non-differentiable; `ws` is unused there too)."""
    E = engine()
    ac = _ac(align_corners)
    out = []
    S = len(cam_flows_fwd)
    if 3 * S <= MAX_JOBS:
        # the 3 warps + 3 error maps of every scale (18 of each per step) as one launch each, then all targets in one launch
        with torch.no_grad():
            B = tgt_img.shape[0]
            cf = [_f32c(t) for t in cam_flows_fwd]
            cb = [_f32c(t) for t in cam_flows_bwd]
            ff = [_f32c(t) for t in flows_fwd]
            hw = [(t.shape[2], t.shape[3]) for t in cf]
            px = [h * w for h, w in hw]
            tgt_l = [pyramid_cache.get(tgt_img, h, w) for h, w in hw]
            rf = [pyramid_cache.get(ref_img_fwd, h, w) for h, w in hw]
            rb = [pyramid_cache.get(ref_img_bwd, h, w) for h, w in hw]
            warped = _carve([B * 3 * px[s] for s in range(S) for _ in range(3)], tgt_img)
            ev = _carve([B * px[s] for s in range(S) for _ in range(6)], tgt_img)       # err x3, valid x3 per scale
            jw, je, jc = _Jobs(), _Jobs(), _Jobs()
            for s in range(S):
                for k, (src, flow) in enumerate(((rf[s], cf[s]), (rb[s], cb[s]), (rf[s], ff[s]))):
                    jw.add([src, flow, warped[3 * s + k]], *hw[s])
                    je.add([tgt_l[s], warped[3 * s + k], ev[6 * s + k], ev[6 * s + 3 + k]], *hw[s])
                target = torch.empty(B, 1, hw[s][0], hw[s][1], device=tgt_img.device, dtype=torch.float32)
                jc.add([ev[6 * s], ev[6 * s + 1], ev[6 * s + 2], ev[6 * s + 3], ev[6 * s + 4], target], *hw[s])
                out.append(target)
            E.call("cc_flow_warp_fwd_jobs", jw.pack(), len(jw), B, 3, 0, ac, STREAM)
            E.call("cc_ssim_err_fwd_jobs", je.pack(), len(je), B, float(wssim), gauss13_ptr(), STREAM)
            E.call("cc_consensus_target_jobs", jc.pack(), len(jc), B, float(wrig), STREAM)
        return out
    with torch.no_grad():
        for i in range(len(cam_flows_fwd)):
            cf, cb, ff = _f32c(cam_flows_fwd[i]), _f32c(cam_flows_bwd[i]), _f32c(flows_fwd[i])
            B, _, h, w = cf.shape
            tgt_s = pyramid_cache.get(tgt_img, h, w)
            rf = pyramid_cache.get(ref_img_fwd, h, w)
            rb = pyramid_cache.get(ref_img_bwd, h, w)
            errs, valids = [], []
            for src, flow in ((rf, cf), (rb, cb), (rf, ff)):
                warped = torch.empty_like(src)
                E.call("cc_flow_warp_fwd", src, flow, warped, B, 3, h, w, 0, ac, STREAM)
                err = torch.empty(B, 1, h, w, device=src.device, dtype=torch.float32)
                valid = torch.empty_like(err)
                E.call("cc_ssim_err_fwd", tgt_s, warped, err, valid, float(wssim), gauss13_ptr(), B, h, w, STREAM)
                errs.append(err)
                valids.append(valid)
            target = torch.empty_like(errs[0])
            E.call("cc_consensus_target", errs[0], errs[1], errs[2], valids[0], valids[1], target, float(wrig),
                   target.numel(), STREAM)
            out.append(target)
    return out


def compute_joint_mask_for_depth(explainability_mask, rigidity_mask_bwd, rigidity_mask_fwd, THRESH):
    """loss_functions.py:204-219 (optional path, broken in train.py -- H10); stock torch."""
    joint_masks = []
    for i in range(len(explainability_mask)):
        e = explainability_mask[i]
        rf = (rigidity_mask_fwd[i] > THRESH).type_as(e)
        rb = (rigidity_mask_bwd[i] > THRESH).type_as(e)
        ej = 1 - (1 - e[:, 1]) * (1 - e[:, 2]).unsqueeze(1) > 0.5
        jf = logical_or(rf.type_as(e), ej.type_as(e)).detach()
        jb = logical_or(rb.type_as(e), ej.type_as(e)).detach()
        joint_masks.append(torch.cat((jb, jb, jf, jf), dim=1))
    return joint_masks


class _ConsensusBCEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cfg, *rest):
        S = cfg.n_scales
        masks = rest[:S]
        cb, cf, tb, tf = (rest[S * (k + 1):S * (k + 2)] for k in range(4))
        E = engine()
        need = ctx.needs_input_grad
        loss_acc = _zeros1(masks[0])
        arena = _GradArena(list(masks), list(need[1:1 + S]))
        if S <= MAX_JOBS:
            B = masks[0].shape[0]
            offs, tot = _partials_for([mk.shape for mk in masks], lambda shp: B)
            part = _empty(tot, masks[0])
            jb = _Jobs()
            for s in range(S):
                e = _f32c(masks[s].detach())
                assert e.shape[1] == 4
                jb.add([e, _f32c(cb[s].detach()), _f32c(cf[s].detach()), _f32c(tb[s].detach()), _f32c(tf[s].detach()),
                        arena.view(s), _off(part, offs[s])], e.shape[2], e.shape[3])
            E.call("cc_consensus_bce_fwd_bwd_jobs", jb.pack(), len(jb), B, part, loss_acc, float(cfg.THRESH), float(cfg.wbce), 1.0,
                   STREAM)
            ctx.arena = arena
            ctx.n_rest = len(rest)
            return loss_acc.reshape(())
        for s in range(S):
            e = _f32c(masks[s].detach())
            B, C, h, w = e.shape
            assert C == 4
            g = arena.view(s)
            nb = E.call("cc_elem_num_blocks", h * w) * B
            E.call("cc_consensus_bce_fwd_bwd", e, _f32c(cb[s].detach()), _f32c(cf[s].detach()), _f32c(tb[s].detach()),
                   _f32c(tf[s].detach()), g, _empty(nb, e), loss_acc, float(cfg.THRESH), float(cfg.wbce), 1.0, B, h, w,
                   STREAM)
        ctx.arena = arena
        ctx.n_rest = len(rest)
        return loss_acc.reshape(())

    @staticmethod
    def backward(ctx, gout):
        g = ctx.arena.scaled(gout)
        return (None,) + tuple(g) + (None,) * (ctx.n_rest - len(g))


def consensus_depth_flow_mask(explainability_mask, census_mask_bwd, census_mask_fwd, exp_masks_bwd_target,
                              exp_masks_fwd_target, THRESH, wbce):
    """loss_functions.py:221-250."""
    assert len(explainability_mask) == len(census_mask_bwd)
    assert len(explainability_mask) == len(census_mask_fwd)
    cfg = _PhotoCfg(n_scales=len(explainability_mask), THRESH=THRESH, wbce=wbce)
    return _ConsensusBCEFn.apply(cfg, *explainability_mask, *census_mask_bwd, *census_mask_fwd, *exp_masks_bwd_target,
                                 *exp_masks_fwd_target)


def weighted_binary_cross_entropy(output, target, weights=None):
    """loss_functions.py:252-261 (public helper; the training path uses the fused kernel above)."""
    if weights is not None:
        assert len(weights) == 2
        loss = weights[1] * (target * torch.log(output + epsilon)) + \
            weights[0] * ((1 - target) * torch.log(1 - output + epsilon))
    else:
        loss = target * torch.log(output + epsilon) + (1 - target) * torch.log(1 - output + epsilon)
    return torch.neg(torch.mean(loss))


# ----------------------------------------------------------------------------- validation metrics
# loss_functions.py:355-467, consumed by train.py's validate_* loops (train.py:588-777).  Device-side torch arithmetic
# on whatever device the inputs live on (they run once per validation batch, far off the training hot path), same
# values as the reference.  The reference returns Python floats through .item() (a host sync per metric); sync=False
# returns 0-dim device tensors instead so that a validation loop can stay asynchronous.
def _upsample_to(pred, size):
    """nn.functional.upsample(pred, size=size, mode='bilinear') (loss_functions.py:359,373,394,414-415): align_corners
    has defaulted to False for upsample since torch 0.4, i.e. for the authors' torch 1.0 as well as today's -- it does
    NOT follow config.align_corners (that knob mirrors grid_sample's later change of default)."""
    return nn.functional.interpolate(pred, size=size, mode='bilinear', align_corners=False)


def _scaled_uv(gt, pred):
    _, _, h_pred, w_pred = pred.size()
    _, _, h_gt, w_gt = gt.size()
    pred = _upsample_to(pred, (h_gt, w_gt))
    return gt[:, 0], gt[:, 1], pred[:, 0] * (w_gt / w_pred), pred[:, 1] * (h_gt / h_pred)


def flow_diff(gt, pred):
    """loss_functions.py:355-365 -> per-pixel end-point error [B,H,W] at the ground truth's resolution."""
    u_gt, v_gt, u_pred, v_pred = _scaled_uv(gt, pred)
    return torch.sqrt(torch.pow((u_gt - u_pred), 2) + torch.pow((v_gt - v_pred), 2))


def compute_epe(gt, pred, sync=True):
    """loss_functions.py:368-388: mean EPE; a third ground-truth channel is a validity mask."""
    bs, nc, h_gt, w_gt = gt.size()
    u_gt, v_gt, u_pred, v_pred = _scaled_uv(gt, pred)
    epe = torch.sqrt(torch.pow((u_gt - u_pred), 2) + torch.pow((v_gt - v_pred), 2))
    if nc == 3:
        valid = gt[:, 2]
        epe = epe * valid
        avg_epe = epe.sum() / (valid.sum() + epsilon)
    else:
        avg_epe = epe.sum() / (bs * h_gt * w_gt)
    avg_epe = avg_epe.detach()
    return avg_epe.item() if sync else avg_epe


def outlier_err(gt, pred, tau=[3, 0.05], sync=True):
    """loss_functions.py:390-409: KITTI Fl outlier ratio (EPE > 3 px AND > 5 % of the flow magnitude)."""
    u_gt, v_gt, u_pred, v_pred = _scaled_uv(gt, pred)
    valid_gt = gt[:, 2]
    epe = torch.sqrt(torch.pow((u_gt - u_pred), 2) + torch.pow((v_gt - v_pred), 2))
    epe = epe * valid_gt
    F_mag = torch.sqrt(torch.pow(u_gt, 2) + torch.pow(v_gt, 2))
    E_0 = (epe > tau[0]).type_as(epe)
    E_1 = ((epe / (F_mag + epsilon)) > tau[1]).type_as(epe)
    n_err = E_0 * E_1 * valid_gt
    f_err = (n_err.sum() / (valid_gt.sum() + epsilon)).detach()
    return f_err.item() if sync else f_err


def compute_all_epes(gt, rigid_pred, non_rigid_pred, rigidity_mask, THRESH=0.5, sync=True):
    """loss_functions.py:411-429 -> [all_epe, rigid_epe, non_rigid_epe, outliers]."""
    _, _, h_pred, w_pred = rigid_pred.size()
    _, _, h_gt, w_gt = gt.size()
    rigidity_pred_mask = _upsample_to(rigidity_mask, (h_pred, w_pred))
    rigidity_gt_mask = _upsample_to(rigidity_mask, (h_gt, w_gt))
    non_rigid_pred = (rigidity_pred_mask <= THRESH).type_as(non_rigid_pred).expand_as(non_rigid_pred) * non_rigid_pred
    rigid_pred = (rigidity_pred_mask > THRESH).type_as(rigid_pred).expand_as(rigid_pred) * rigid_pred
    total_pred = non_rigid_pred + rigid_pred
    gt_non_rigid = (rigidity_gt_mask <= THRESH).type_as(gt).expand_as(gt) * gt
    gt_rigid = (rigidity_gt_mask > THRESH).type_as(gt).expand_as(gt) * gt
    all_epe = compute_epe(gt, total_pred, sync)
    rigid_epe = compute_epe(gt_rigid, rigid_pred, sync)
    non_rigid_epe = compute_epe(gt_non_rigid, non_rigid_pred, sync)
    outliers = outlier_err(gt, total_pred, sync=sync)
    return [all_epe, rigid_epe, non_rigid_epe, outliers]


def compute_errors(gt, pred, crop=True):
    """loss_functions.py:432-467: [abs_diff, abs_rel, sq_rel, a1, a2, a3] of median-scaled depth (Garg/Eigen crop).
    gt, pred: [B,H,W]; returns 0-dim tensors like the reference (no host sync inside)."""
    abs_diff, abs_rel, sq_rel, a1, a2, a3 = 0, 0, 0, 0, 0, 0
    batch_size = gt.size(0)
    if crop:
        crop_mask = torch.zeros_like(gt[0], dtype=torch.bool)               # gt[0] != gt[0], :441
        y1, y2 = int(0.40810811 * gt.size(1)), int(0.99189189 * gt.size(1))
        x1, x2 = int(0.03594771 * gt.size(2)), int(0.96405229 * gt.size(2))
        crop_mask[y1:y2, x1:x2] = True
    for current_gt, current_pred in zip(gt, pred):
        valid = (current_gt > 0) & (current_gt < 80)
        if crop:
            valid = valid & crop_mask
        valid_gt = current_gt[valid]
        valid_pred = current_pred[valid].clamp(1e-3, 80)
        valid_pred = valid_pred * torch.median(valid_gt) / torch.median(valid_pred)
        thresh = torch.max((valid_gt / valid_pred), (valid_pred / valid_gt))
        a1 += (thresh < 1.25).float().mean()
        a2 += (thresh < 1.25 ** 2).float().mean()
        a3 += (thresh < 1.25 ** 3).float().mean()
        abs_diff += torch.mean(torch.abs(valid_gt - valid_pred))
        abs_rel += torch.mean(torch.abs(valid_gt - valid_pred) / valid_gt)
        sq_rel += torch.mean(((valid_gt - valid_pred) ** 2) / valid_gt)
    return [metric / batch_size for metric in [abs_diff, abs_rel, sq_rel, a1, a2, a3]]


def edge_aware_smoothness_per_pixel(img, pred):
    """loss_functions.py:263-284 (the reference drops into ipdb before returning, :283; the value is what it returns)."""
    def gradient_x(t):
        return t[:, :, :-1, :] - t[:, :, 1:, :]

    def gradient_y(t):
        return t[:, :, :, :-1] - t[:, :, :, 1:]
    weights_x = torch.exp(-torch.mean(torch.abs(gradient_x(img)), 1, keepdim=True))
    weights_y = torch.exp(-torch.mean(torch.abs(gradient_y(img)), 1, keepdim=True))
    smoothness_x = torch.abs(gradient_x(pred)) * weights_x
    smoothness_y = torch.abs(gradient_y(pred)) * weights_y
    return smoothness_x + smoothness_y                                     # broadcasting error for H != W, as upstream
