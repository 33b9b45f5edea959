"""Planned backward of the four CC networks: a define-by-run tape over the engine's kernels.

torch.autograd sees one node per NETWORK (``run_network``); inside, the forward pass is recorded on this tape and the
backward pass is scheduled by hand, which removes what a per-layer autograd graph cannot avoid
(profiles/r02_step_trace_fin.txt: ~135 accumulation adds, 22 torch.cat copies, ~100 activation-backward passes per step):

  * gradients of fan-out tensors are accumulated by their CONSUMERS -- every data-gradient launch either writes its result,
    adds it to what is already there (epilogue operand ``add`` aliasing the output), or picks up another tensor as ``add``
    (the gradient a residual shortcut carries is never copied or added by a launch of its own);
  * the last contributor of a tensor that came out of an activation multiplies act'(y) into the same epilogue,
    gx = (sum + add) * act'(y) (cc_conv2d_dgrad_group_add), so the producer needs no activation-backward pass -- for a
    ResNet block (models/DispResNet6.py:31-43) that is one launch instead of data-gradient + add + ReLU backward;
  * concatenations are buffers the producers write their channel slice of (batch-stride arguments of the kernels); the
    gradient of a concatenation is read in place by the producers' backward.

Semantics are those of the per-layer path (cc_amd/ops.py, kept for the alternative architectures): same kernels, same
arithmetic per element; only the order in which gradient contributions are summed can differ.
"""
import ctypes

import torch

from . import config, ops
from ._lib import engine, STREAM

ACT = ops.ACT


def _bs(t):
    return t.stride(0)


def _new(shape, ref):
    return torch.empty(shape, device=ref.device, dtype=torch.float32)


def _dense(t):
    """contiguous, or a per-image dense channel slice of a wider NCHW tensor (read / written through batch strides)"""
    B, C, H, W = t.shape
    st = t.stride()
    return st[3] == 1 and st[2] == W and st[1] == H * W and (B == 1 or st[0] >= C * H * W)



class BnCounters:
    """``num_batches_tracked`` of every BatchNorm layer of the trainer's networks as views of ONE int64 buffer: the step bumps the
    counters of the layers that ran in training mode with one add instead of one launch per layer (13 in DispResNet6).  Active
    only while the trainer runs a step (tape.BN_COUNTERS); a network called on its own counts per layer as torch does."""

    def __init__(self, nets):
        self.mods = [m for n in nets if n is not None for m in n.modules()
                     if getattr(m, "num_batches_tracked", None) is not None and hasattr(m, "running_mean")]
        self.index = {id(m): i for i, m in enumerate(self.mods)}
        self.ran = set()
        self.buf = None
        self.inc = {}
        if self.mods:
            self._adopt()

    def _adopt(self):
        self.buf = torch.stack([m.num_batches_tracked.detach().reshape(()) for m in self.mods]).clone()
        self.inc = {}
        for i, m in enumerate(self.mods):
            m._buffers["num_batches_tracked"] = self.buf[i]

    def begin(self):
        self.ran = set()
        # a module moved / re-created its buffers since (net.to(...), load with assign=True): adopt the new values
        if self.buf is not None and any(m.num_batches_tracked.data_ptr() != self.buf[i].data_ptr() for i, m in enumerate(self.mods)):
            self._adopt()

    def commit(self):
        """one add for the layers that ran (the increment vector is cached per set: built outside stream captures, in the
        trainer's eager warm-up steps)"""
        if self.buf is None or not self.ran:
            return
        key = tuple(sorted(self.index[k] for k in self.ran))
        if key not in self.inc:
            v = torch.zeros(len(self.mods), dtype=self.buf.dtype)
            v[list(key)] = 1
            self.inc[key] = v.to(self.buf.device)
        self.buf.add_(self.inc[key])


BN_COUNTERS = None

# Set by CCTrainer for the duration of a pipelined step: called with the network module at the END of that network's backward
# pass (inside its autograd node, i.e. on the network's stream, behind its last gradient launch) -- the trainer then runs the
# network's gradient exchange, Adam segment and weight-image refresh right there instead of behind the whole backward pass.
NET_DONE = None
# ... and with (module, tag) when the backward pass comes back to a point the network's forward marked (Tape.mark): every parameter
# used AFTER that point has its complete gradient (parked weight-gradient launches aside: the callee flushes them), so the trainer
# can start exchanging / updating that part of the network's bucket segment while the rest of its backward pass still runs.
NET_MARK = None
# diagnosis (tools/grad_pin.py): a dict -> every tensor a network's forward tagged with Tape.tap() leaves its accumulated gradient
# there when its producer picks it up in the backward pass: name -> (gradient clone or None, already-pre-activation flag)
TAPS = None

class TT:
    """A tensor on the tape + the state of its gradient during the backward pass."""
    __slots__ = ("t", "act", "act_a", "act_b", "uses", "remaining", "grad", "pend", "pre", "needs", "tapname")

    def __init__(self, t, needs=True, act=0, act_a=1.0, act_b=0.0):
        assert t.dtype == torch.float32 and (t.dim() != 4 or _dense(t)), (tuple(t.shape), t.stride())
        self.t = t
        self.act, self.act_a, self.act_b = act, act_a, act_b      # activation the producer applied (its backward can be deferred)
        self.uses = 0            # recorded consumers
        self.remaining = 0       # contributions still to come (backward pass)
        self.grad = None         # accumulated gradient (own buffer)
        self.pend = []           # other tensors that are additive parts of the gradient (not summed yet)
        self.pre = False         # grad is already w.r.t. the producer's PRE-activation
        self.needs = needs       # gradient wanted at all
        self.tapname = None      # diagnosis: see TAPS

    # ---- contributions (called by the consumers' backward, in reverse forward order)
    def skip(self):
        self.remaining -= 1

    def add_alias(self, g):
        """the tensor g is an additive part of this gradient (nothing is launched)"""
        self.remaining -= 1
        if self.needs and g is not None:
            self.pend.append(g)

    def _fold_extra_pend(self):
        """at most one pending alias can ride in an epilogue: sum the others into an own buffer -- one cc_sum_strided launch per
        eight parts (the parts are channel slices of concat gradients with their own batch strides)"""
        if not (len(self.pend) > 1 or (self.pend and self.grad is not None)):
            return
        parts = self.pend
        self.pend = []
        if self.t.dim() == 4 and all(p.dim() == 4 and p.dtype == torch.float32 and _dense(p) and p.shape == self.t.shape for p in parts) \
                and (self.grad is None or _dense(self.grad)) and not ops._dbg.no_sum_n:
            acc = self.grad is not None
            if not acc:
                self.grad = _new(self.t.shape, self.t)
            B, C, H, W = self.t.shape
            for c0 in range(0, len(parts), 8):
                ch = parts[c0:c0 + 8]
                src = (ctypes.c_long * len(ch))(*[p.data_ptr() for p in ch])
                sbs = (ctypes.c_long * len(ch))(*[p.stride(0) if B > 1 else C * H * W for p in ch])
                engine().call("cc_sum_strided", len(ch), ctypes.addressof(src), ctypes.addressof(sbs), self.grad,
                              self.grad.stride(0) if B > 1 else C * H * W, B, C * H * W, int(acc), STREAM)
                acc = True
            return                      # (stream-ordered: the parts may be released right after the launch is queued)
        while parts:
            g = parts.pop()
            if self.grad is None:
                self.grad = (parts.pop() + g) if parts else g.clone()
            else:
                self.grad.add_(g)

    def from_conv(self, launch, can_mul=True):
        """launch(gx, add, mul): a data-gradient kernel whose epilogue writes gx = (sum + add) [* act'(mul)]."""
        self.remaining -= 1
        if not self.needs:
            return
        self._fold_extra_pend()
        add = self.grad if self.grad is not None else (self.pend.pop() if self.pend else None)
        last = can_mul and self.remaining == 0 and self.act != 0
        if self.grad is None:
            self.grad = _new(self.t.shape, self.t)
        launch(self.grad, add, self.t if last else None)
        if last:
            self.pre = True

    def from_writer(self, write):
        """write(gx, accumulate): any other backward kernel."""
        self.remaining -= 1
        if not self.needs:
            return
        if self.grad is None and self.pend:
            self._fold_extra_pend()
            if self.pend:
                self.grad = _new(self.t.shape, self.t)
                self.grad.copy_(self.pend.pop())
        if self.grad is None:
            self.grad = _new(self.t.shape, self.t)
            write(self.grad, False)
        else:
            write(self.grad, True)

    def final(self):
        """-> (gradient or None, already-pre-activation flag); called by the producer's backward."""
        g, pre = self._final()
        if self.tapname is not None and TAPS is not None:
            TAPS[self.tapname] = (g.detach().clone() if g is not None else None, pre)
        return g, pre

    def _final(self):
        if not self.needs:
            return None, False
        if self.grad is None:
            if not self.pend:
                return None, False
            if len(self.pend) == 1:
                return self.pend[0], False
        self._fold_extra_pend()
        return self.grad, self.pre


class ConcatBuffer:
    """torch.cat((p0, p1, ...), 1) as ONE buffer the producers write their channel slice of."""

    def __init__(self, tape, B, chans, H, W, ref):
        self.tape = tape
        self.buf = _new((B, sum(chans), H, W), ref)
        self.offs = [0]
        for c in chans:
            self.offs.append(self.offs[-1] + c)
        self.parts = [None] * len(chans)

    def slot(self, i):
        return self.buf[:, self.offs[i]:self.offs[i + 1]]

    def put(self, i, tt):
        """part i: a TT produced in place (its .t IS slot(i)) or elsewhere (copied in: one launch, as torch.cat does)"""
        if tt.t.data_ptr() != self.slot(i).data_ptr() or tt.t.stride() != self.slot(i).stride():
            assert tuple(tt.t.shape) == tuple(self.slot(i).shape), (tuple(tt.t.shape), tuple(self.slot(i).shape))
            self.slot(i).copy_(tt.t)
        self.parts[i] = tt

    def done(self):
        return self.tape._cat(self)


class Tape:
    def __init__(self, record=True):
        self.record = record
        self.nodes = []           # backward closures in forward order
        self.tts = []
        self.param_grads = {}     # id(param) -> (param, gradient) for parameters without a flat-bucket sink
        self.owner = None         # the network module (for NET_MARK)
        self.E = engine()

    # ------------------------------------------------------------------ bookkeeping
    def leaf(self, t, needs=False):
        tt = TT(t if t.is_contiguous() or _dense(t) else t.contiguous(), needs=needs and self.record)
        self.tts.append(tt)
        return tt

    def _tt(self, t, **kw):
        tt = TT(t, needs=self.record, **kw)
        self.tts.append(tt)
        return tt

    def _use(self, *xs):
        for x in xs:
            if x is not None:
                x.uses += 1

    def _node(self, fn):
        if self.record:
            self.nodes.append(fn)

    def _param_grad(self, p, shape=None):
        """-> (gradient buffer, accumulate flag) for parameter p: the optimizer's flat bucket when the trainer registered one"""
        sink = ops._sink(p)
        if sink is not None:
            return sink, True
        ent = self.param_grads.get(id(p))
        if ent is not None:
            return ent[1], True
        g = torch.empty_like(p)
        self.param_grads[id(p)] = (p, g)
        return g, False

    def concat(self, B, chans, H, W, ref):
        return ConcatBuffer(self, B, chans, H, W, ref)

    def tap(self, tt, name):
        """diagnosis: leave tt's accumulated gradient in TAPS[name] during the backward pass (no-op unless TAPS is a dict)"""
        if TAPS is not None:
            tt.tapname = name
        return tt

    def mark(self, tag):
        """A point of the forward pass: in the backward pass NET_MARK(owner module, tag) is called once everything recorded after
        this point has been differentiated (see NET_MARK)."""
        if self.record:
            owner = self.owner

            def bwd():
                if NET_MARK is not None:
                    NET_MARK(owner, tag)
            self.nodes.append(bwd)

    def _cat(self, cb):
        assert all(p is not None for p in cb.parts)
        out = self._tt(cb.buf)
        self._use(*cb.parts)
        parts, offs = cb.parts, cb.offs

        def bwd():
            g, _ = out.final()
            for i, p in enumerate(parts):
                if g is None:
                    p.skip()
                else:
                    p.add_alias(g[:, offs[i]:offs[i + 1]])
        self._node(bwd)
        return out

    # ------------------------------------------------------------------ activation backward + bias gradient
    def _act_bias(self, gs, ys, act, act_a, act_b, biases, ref, targets=None):
        """geff_k = g_k * act'(y_k) (act 0: geff_k = g_k, no pass unless a bias gradient is wanted); bias gradients summed
        over (B, H, W).  -> list of geff.  targets: the (gradient buffer, accumulate) pairs of the biases when the caller has
        resolved them already -- _param_grad hands out a FRESH buffer with accumulate = False exactly once per parameter, so the
        per-member calls below must not ask again (they would be told to accumulate into the uninitialised buffer)."""
        G = len(gs)
        B, C, H, W = gs[0].shape
        want_b = biases is not None and biases[0] is not None and biases[0].requires_grad
        if act == 0 and not want_b:
            return list(gs)
        gbs, acc, tgt = [None] * G, False, None
        if want_b:
            tgt = targets if targets is not None else [self._param_grad(b) for b in biases]
            acc = all(a for _, a in tgt)
            if not acc and any(a for _, a in tgt):       # mixed: first use of some, second use of others -> one by one
                out = []
                for k in range(G):
                    out += self._act_bias([gs[k]], [ys[k]], act, act_a, act_b, [biases[k]], ref, targets=[tgt[k]])
                return out
            gbs = [t for t, _ in tgt]
        geffs = [_new((B, C, H, W), ref) for _ in range(G)] if act != 0 else [None] * G
        if act == 0 and acc and config.bias_table and ops.wgrad_queue.enabled and not ops._dbg.no_wgrad_defer:
            # pure bias sums inside a trainer stage: parked, all layers of the stage in one launch (cc_bias_grad_table)
            for g, gb in zip(gs, gbs):
                ops.wgrad_reduces.park_bias(g, gb, B, C, H, W, _bs(g))
            return list(gs)
        gy_bs = _bs(gs[0])
        uniform = all(_bs(g) == gy_bs for g in gs) and (act == 0 or all(_bs(y) == _bs(ys[0]) for y in ys))
        if not uniform:
            out = []
            for k in range(G):
                out += self._act_bias([gs[k]], [ys[k]], act, act_a, act_b, [biases[k]] if want_b else None, ref,
                                      targets=[tgt[k]] if want_b else None)
            return out
        for c0 in range(0, G, 4):
            sl = slice(c0, c0 + 4)
            ops_gys, ops_ys, ops_ge, ops_gb = gs[sl], (ys[sl] if act != 0 else [None] * len(gs[sl])), geffs[sl], gbs[sl]
            n = len(ops_gys)
            a1, a2, a3, a4 = ops._parr(ops_gys), ops._parr(ops_ys), ops._parr(ops_ge), ops._parr(ops_gb)
            ws = ops._ws(self.E.call("cc_act_bwd_ws_bytes", C) * n, ref)
            args = (n, ops._addr(a1), ops._addr(a2) if act != 0 else 0, ops._addr(a3) if act != 0 else 0,
                    ops._addr(a4) if want_b else 0, ws, B, C, H, W, gy_bs, _bs(ops_ys[0]) if act != 0 else C * H * W, C * H * W,
                    act, act_a, act_b, int(acc))
            if want_b and acc and ops.wgrad_queue.enabled and not ops._dbg.no_wgrad_defer:
                ptrs = [t.data_ptr() for t in ops_gb]
                if ops.wgrad_reduces.targets.intersection(ptrs) or len(set(ptrs)) != len(ptrs):
                    ops.wgrad_reduces._cur().flush()
                ops.wgrad_reduces.targets.update(ptrs)
                red = (ctypes.c_long * (16 * n))()
                nred = ctypes.c_int(0)
                self.E.call("cc_act_bwd_bias_group_defer", *args, ctypes.addressof(red), n, ctypes.addressof(nred), STREAM)
                if nred.value:
                    ops.wgrad_reduces.keep.append((ws, ops_gb, ops_gys))
                    ops.wgrad_reduces.desc.extend(red[:16 * nred.value])
            else:
                self.E.call("cc_act_bwd_bias_group", *args, STREAM)
        return geffs if act != 0 else list(gs)

    # ------------------------------------------------------------------ weight gradients
    def _wgrad(self, a_list, x_list, w_list, geom):
        """gw_k (+)= a_k (*) x_k for G same-shaped problems; geom = (B, M, AH, AW, Cin, IH, IW, R, S, si, pad, o_sm, o_sc)."""
        B, M, AH, AW, Cin, IH, IW, R, S, si, pad, o_sm, o_sc = geom
        live = [k for k, w in enumerate(w_list) if w.requires_grad]
        if not live:
            return
        tg = [self._param_grad(w_list[k]) for k in live]
        a_bs, x_bs = _bs(a_list[live[0]]), _bs(x_list[live[0]])
        uniform = all(_bs(a_list[k]) == a_bs and _bs(x_list[k]) == x_bs for k in live)
        if ops.wgrad_queue.enabled and all(acc for _, acc in tg) and all(ops._sink(w_list[k]) is not None for k in live):
            for k, (gw, _) in zip(live, tg):
                ops.wgrad_queue.push(geom + (_bs(a_list[k]), _bs(x_list[k])), a_list[k], x_list[k], gw)
            return
        groups = [live] if (uniform and len({acc for _, acc in tg}) == 1) else [[k] for k in live]
        acc_of = {k: acc for k, (_, acc) in zip(live, tg)}
        gw_of = {k: gw for k, (gw, _) in zip(live, tg)}
        for grp in groups:
            for c0 in range(0, len(grp), 4):
                ch = grp[c0:c0 + 4]
                ops._wgrad_group([a_list[k] for k in ch], [x_list[k] for k in ch], [gw_of[k] for k in ch], x_list[ch[0]], B, M, AH, AW,
                                 _bs(a_list[ch[0]]), Cin, IH, IW, _bs(x_list[ch[0]]), R, S, si, pad, o_sm, o_sc, int(acc_of[ch[0]]))

    # ------------------------------------------------------------------ convolution (G same-shaped problems per launch)
    def conv(self, x, w, b, stride, pad, act=None, act_a=1.0, act_b=0.0, residual=None, out=None):
        return self.conv_group([x], [w], [b], stride, pad, act, act_a, act_b, [residual] if residual is not None else None,
                               [out] if out is not None else None)[0]

    def conv_group(self, xs, ws_, bs, stride, pad, act=None, act_a=1.0, act_b=0.0, residuals=None, outs=None):
        """[act(conv2d(x_k, w_k) + b_k + r_k)]: nn.Conv2d with the activation (and a residual add) in the epilogue.
        outs: optional destination views (channel slices of concat buffers)."""
        E = self.E
        G = len(xs)
        act = ACT[act] if not isinstance(act, int) else act
        has_bias = bs is not None and bs[0] is not None
        B, Cin, IH, IW = xs[0].t.shape
        Cout, _, R, S = ws_[0].shape
        OH = (IH + 2 * pad - R) // stride + 1
        OW = (IW + 2 * pad - S) // stride + 1
        ys = [o if o is not None else None for o in outs] if outs is not None else [None] * G
        ys = [y if y is not None else _new((B, Cout, OH, OW), xs[0].t) for y in ys]
        x_bs, y_bs = _bs(xs[0].t), _bs(ys[0])
        rs = [r.t for r in residuals] if residuals is not None else None
        res_bs = _bs(rs[0]) if rs is not None else 0
        uniform = all(_bs(x.t) == x_bs for x in xs) and all(_bs(y) == y_bs for y in ys) and (rs is None or all(_bs(r) == res_bs for r in rs))
        geom = (B, Cin, IH, IW, Cout, R, S, stride, pad, OH, OW)

        def launch(idx):
            n = len(idx)
            wsb = ops._ws(E.call("cc_conv2d_fwd_group_ws_bytes", n, *geom), xs[0].t)
            pks = [ops.packs.get("fwd", ws_[k], geom) for k in idx]
            ax, aw, ab = ops._parr([xs[k].t for k in idx]), ops._parr([ws_[k] for k in idx]), ops._parr([bs[k] if has_bias else None for k in idx])
            ar = ops._parr([rs[k] for k in idx]) if rs is not None else None
            ay, ap = ops._parr([ys[k] for k in idx]), ops._parr(pks)
            E.call("cc_conv2d_fwd_group", n, ops._addr(ax), ops._addr(aw), ops._addr(ab), ops._addr(ar) if ar is not None else 0,
                   ops._addr(ay), wsb, ops._addr(ap), B, Cin, IH, IW, _bs(xs[idx[0]].t), Cout, R, S, stride, pad, OH, OW, _bs(ys[idx[0]]),
                   _bs(rs[idx[0]]) if rs is not None else 0, act, float(act_a), float(act_b), STREAM)
        if uniform:
            launch(list(range(G)))
        else:
            for k in range(G):
                launch([k])
        youts = [self._tt(y, act=act, act_a=float(act_a), act_b=float(act_b)) for y in ys]
        self._use(*xs)
        if residuals is not None:
            self._use(*residuals)
        if not self.record:
            return youts
        xs_, res_ = list(xs), (list(residuals) if residuals is not None else None)

        def bwd():
            fin = [y.final() for y in youts]
            live = [k for k in range(G) if fin[k][0] is not None]
            for k in range(G):
                if k not in live:
                    xs_[k].skip()
                    if res_ is not None:
                        res_[k].skip()
            if not live:
                return
            # activation backward (unless the consumers already applied it) + bias gradients
            gz = {}
            for pre in (False, True):
                idx = [k for k in live if fin[k][1] == pre]
                if idx:
                    ge = self._act_bias([fin[k][0] for k in idx], [youts[k].t for k in idx], 0 if pre else act, act_a, act_b,
                                        [bs[k] for k in idx] if has_bias else None, xs_[0].t)
                    for k, g in zip(idx, ge):
                        gz[k] = g
            if res_ is not None:
                for k in live:
                    res_[k].add_alias(gz[k])
            # data gradients: grouped when the epilogue operands agree, else one by one
            dgeom = (B, Cout, OH, OW, Cin, R, S, stride, pad, IH, IW)
            plan = {}
            for k in live:
                xs_[k].from_conv(lambda gx, add, mul, k=k: plan.__setitem__(k, (gx, add, mul)))
            todo = [k for k in live if k in plan]
            groups, last = [], {}
            for k in todo:
                sg = (plan[k][1] is not None, plan[k][2] is not None, _bs(gz[k]), _bs(plan[k][0]),
                      _bs(plan[k][1]) if plan[k][1] is not None else 0, _bs(xs_[k].t), xs_[k].act, xs_[k].act_a, xs_[k].act_b)
                # One launch only for problems with the same epilogue form AND distinct targets; and the launches run in creation
                # order, so a problem may only join a launch that comes AFTER the one holding the previous write of its target (x
                # used twice: the second contribution accumulates onto the first -- its `add` aliases the target -- and must not be
                # pulled into an earlier launch of its epilogue form, where it would read the target before it is written)
                tgt = plan[k][0].data_ptr()
                first = last.get(tgt, -1) + 1
                for gi in range(first, len(groups)):
                    if groups[gi][0] == sg:
                        groups[gi][1].append(k)
                        last[tgt] = gi
                        break
                else:
                    groups.append((sg, [k]))
                    last[tgt] = len(groups) - 1
            for sg, idx in groups:
                n = len(idx)
                has_add, has_mul = sg[0], sg[1]
                wsb = ops._ws(E.call("cc_conv2d_dgrad_group_ws_bytes", n, *dgeom), xs_[0].t)
                pks = [ops.packs.get("dgrad", ws_[k], dgeom + (Cin * R * S, R * S)) for k in idx]
                a1, a2, a3 = ops._parr([gz[k] for k in idx]), ops._parr([ws_[k] for k in idx]), ops._parr([plan[k][0] for k in idx])
                a4 = ops._parr([plan[k][2] for k in idx]) if has_mul else None
                a5 = ops._parr([plan[k][1] for k in idx]) if has_add else None
                a6 = ops._parr(pks)
                E.call("cc_conv2d_dgrad_group_add", n, ops._addr(a1), ops._addr(a2), ops._addr(a3), ops._addr(a4) if has_mul else 0,
                       ops._addr(a5) if has_add else 0, wsb, ops._addr(a6), B, Cout, OH, OW, sg[2], Cin, R, S, stride, pad, IH, IW,
                       sg[3], sg[5] if has_mul else 0, sg[4], Cin * R * S, R * S, sg[6] if has_mul else 0,
                       sg[7] if has_mul else 1.0, sg[8] if has_mul else 0.0, STREAM)
            self._wgrad([gz[k] for k in live], [xs_[k].t for k in live], [ws_[k] for k in live],
                        (B, Cout, OH, OW, Cin, IH, IW, R, S, stride, pad, Cin * R * S, R * S))

        self._node(bwd)
        return youts

    # ------------------------------------------------------------------ transposed convolution
    def conv_transpose(self, x, w, b, stride, pad, out_pad, act=None, act_b=0.0, out=None):
        """act(conv_transpose2d(x, w) + b) (nn.ConvTranspose2d): the transposed arithmetic of the data-gradient kernel."""
        E = self.E
        act = ACT[act] if not isinstance(act, int) else act
        B, Cin, IH, IW = x.t.shape
        _, Cout, R, S = w.shape
        OH = (IH - 1) * stride - 2 * pad + R + out_pad
        OW = (IW - 1) * stride - 2 * pad + S + out_pad
        y = out if out is not None else _new((B, Cout, OH, OW), x.t)
        tgeom = (B, Cin, IH, IW, Cout, R, S, stride, pad, OH, OW)
        wsb = ops._ws(E.call("cc_conv2d_dgrad_ws_bytes", *tgeom), x.t)
        pk = ops.packs.get("dgrad", w, tgeom + (Cout * R * S, R * S))
        E.call("cc_conv2d_dgrad", x.t, w, b, y, wsb, pk, B, Cin, IH, IW, _bs(x.t), Cout, R, S, stride, pad, OH, OW, _bs(y),
               Cout * R * S, R * S, act, 1.0, float(act_b), STREAM)
        yt = self._tt(y, act=act, act_a=1.0, act_b=float(act_b))
        self._use(x)
        if not self.record:
            return yt

        def bwd():
            g, pre = yt.final()
            if g is None:
                x.skip()
                return
            gz = self._act_bias([g], [yt.t], 0 if pre else act, 1.0, act_b, [b] if b is not None else None, x.t)[0]
            # d/dx of a transposed conv is a plain strided conv of gz; the [Cin,Cout,R,S] weight IS its [M,C,R,S] weight

            def launch(gx, add, mul):
                fgeom = (B, Cout, OH, OW, Cin, R, S, stride, pad, IH, IW)
                ws2 = ops._ws(E.call("cc_conv2d_fwd_ws_bytes", *fgeom), x.t)
                pk2 = ops.packs.get("fwd", w, fgeom)
                E.call("cc_conv2d_fwd", gz, w, None, add, gx, ws2, pk2, B, Cout, OH, OW, _bs(gz), Cin, R, S, stride, pad, IH, IW, _bs(gx),
                       _bs(add) if add is not None else 0, 0, 1.0, 0.0, STREAM)
                assert mul is None
            # (the forward-arithmetic entry has no act'(mul) epilogue: let a later contributor of x apply it)
            x.from_conv(launch, can_mul=False)
            self._wgrad([x.t], [gz], [w], (B, Cin, IH, IW, Cout, OH, OW, R, S, stride, pad, Cout * R * S, R * S))
        self._node(bwd)
        return yt

    # ------------------------------------------------------------------ batch norm (training mode), up-sampling
    def batch_norm(self, x, mod):
        E = self.E
        if not mod.training:
            y = ops.batch_norm(x.t, mod.weight, mod.bias, mod.running_mean, mod.running_var, None, False, mod.momentum, mod.eps)
            yt = self._tt(y)
            self._use(x)
            if self.record:
                def bwd_eval():
                    g, _ = yt.final()
                    if g is None:
                        x.skip()
                        return
                    B, C, H, W = g.shape
                    # affine parameters that still train in eval mode (not the reference's use, but nn.BatchNorm2d allows it)
                    want_w = mod.weight is not None and mod.weight.requires_grad
                    want_b = mod.bias is not None and mod.bias.requires_grad
                    if want_w or want_b:
                        gw, gb = ops.bn_eval_affine_grads(g.contiguous(), x.t, mod.running_mean, mod.running_var, mod.eps, want_w, want_b)
                        for p_, gp in ((mod.weight, gw), (mod.bias, gb)):
                            if gp is not None:
                                buf, acc = self._param_grad(p_)
                                buf.add_(gp) if acc else buf.copy_(gp)

                    def write(gx, accumulate):
                        dst = gx if not accumulate else torch.empty_like(gx)
                        E.call("cc_bn_eval_fwd", g.contiguous(), mod.weight, None, mod.running_mean, mod.running_var, dst, ops._ws(8 * C, g),
                               B, C, H, W, float(mod.eps), 1, STREAM)
                        if accumulate:
                            gx.add_(dst)
                    x.from_writer(write)
                self._node(bwd_eval)
            return yt
        xt = x.t if x.t.is_contiguous() else x.t.contiguous()
        B, C, H, W = xt.shape
        if B * H * W == 1:
            raise ValueError("Expected more than 1 value per channel when training, got input size %s" % (list(xt.shape),))
        if mod.num_batches_tracked is not None:
            if BN_COUNTERS is not None and id(mod) in BN_COUNTERS.index:
                BN_COUNTERS.ran.add(id(mod))        # the trainer bumps all counters of the step with one add
            else:
                mod.num_batches_tracked.add_(1)
        y = torch.empty_like(xt)
        mean = torch.empty(C, device=xt.device, dtype=torch.float32)
        invstd = torch.empty_like(mean)
        E.call("cc_bn_train_fwd", xt, mod.weight, mod.bias, mod.running_mean, mod.running_var, y, mean, invstd,
               ops._ws(E.call("cc_bn_ws_bytes", C), xt), B, C, H, W, float(mod.momentum), float(mod.eps), STREAM)
        yt = self._tt(y)
        self._use(x)
        if not self.record:
            return yt

        def bwd():
            g, _ = yt.final()
            if g is None:
                x.skip()
                return
            g = g if g.is_contiguous() else g.contiguous()
            wg = mod.weight is not None and mod.weight.requires_grad
            bg = mod.bias is not None and mod.bias.requires_grad
            gw, acc_w = self._param_grad(mod.weight) if wg else (None, False)
            gb, acc_b = self._param_grad(mod.bias) if bg else (None, False)
            assert not (wg and bg) or acc_w == acc_b

            def write(gx, accumulate):
                dst = gx if not accumulate else torch.empty_like(gx)
                E.call("cc_bn_train_bwd", g, xt, mod.weight, mean, invstd, dst, gw, gb, ops._ws(E.call("cc_bn_ws_bytes", C), xt),
                       B, C, H, W, int(acc_w or acc_b), STREAM)
                if accumulate:
                    gx.add_(dst)
            if x.needs:
                x.from_writer(write)
            else:
                x.skip()
                write(torch.empty_like(xt), False)
        self._node(bwd)
        return yt

    def upsample2x(self, x, scale=1.0, out=None):
        """scale * F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=False) in one launch."""
        E = self.E
        B, C, H, W = x.t.shape
        y = out if out is not None else _new((B, C, 2 * H, 2 * W), x.t)
        E.call("cc_upsample2x_fwd", x.t, y, B, C, H, W, _bs(x.t), _bs(y), float(scale), STREAM)
        yt = self._tt(y)
        self._use(x)
        if self.record:
            def bwd():
                g, _ = yt.final()
                if g is None:
                    x.skip()
                    return
                x.from_writer(lambda gx, acc: E.call("cc_upsample2x_bwd", g, gx, B, C, H, W, _bs(g), _bs(gx), float(scale), int(acc), STREAM))
            self._node(bwd)
        return yt

    # ------------------------------------------------------------------ cost volume, feature warp (Back2Future)
    def corr_pair(self, a, b, c, inv_b, inv_c, out=None):
        """cat(correlate(a, b)[:, perm_b], correlate(a, c)[:, perm_c]) written at channels 0..161 of `out`'s buffer."""
        E = self.E
        B, C, H, W = a.t.shape
        at, bt, ct = [t.t if t.t.is_contiguous() else t.t.contiguous() for t in (a, b, c)]
        if out is None:
            out = _new((B, 162, H, W), at)
        # the kernels address out[n, off + ch] as base + (n * total + off + ch) * H * W: passing the SLICE's first element as
        # base with the wide buffer's channel count addresses a channel slice of a concat buffer in place
        tot = _bs(out) // (H * W)
        E.call("cc_corr9x9_fwd", at, bt, out, inv_b, B, C, H, W, tot, 0, STREAM)
        E.call("cc_corr9x9_fwd", at, ct, out, inv_c, B, C, H, W, tot, 81, STREAM)
        yt = self._tt(out)
        self._use(a, b, c)
        if self.record:
            def bwd():
                g, _ = yt.final()
                if g is None:
                    a.skip(), b.skip(), c.skip()
                    return
                gtot = _bs(g) // (H * W)
                ga, gb_, gc = torch.empty_like(at), torch.empty_like(bt), torch.empty_like(ct)
                E.call("cc_corr9x9_bwd", g, at, bt, ga, gb_ if b.needs else None, inv_b, B, C, H, W, gtot, 0, 0, STREAM)
                E.call("cc_corr9x9_bwd", g, at, ct, ga, gc if c.needs else None, inv_c, B, C, H, W, gtot, 81, 1, STREAM)
                a.add_alias(ga)
                b.add_alias(gb_) if b.needs else b.skip()
                c.add_alias(gc) if c.needs else c.skip()
            self._node(bwd)
        return yt

    def feature_warp(self, x, flow, flow_scale, align_corners):
        """models/back2future.py:287-321 Model.warp (border padding) of x by flow * flow_scale."""
        E = self.E
        xt = x.t if x.t.is_contiguous() else x.t.contiguous()
        ft = flow.t if flow.t.is_contiguous() else flow.t.contiguous()
        B, C, H, W = xt.shape
        y = torch.empty_like(xt)
        E.call("cc_feature_warp_fwd", xt, ft, y, B, C, H, W, int(align_corners), float(flow_scale), STREAM)
        yt = self._tt(y)
        self._use(x, flow)
        if self.record:
            def bwd():
                g, _ = yt.final()
                if g is None:
                    x.skip(), flow.skip()
                    return
                g = g if g.is_contiguous() else g.contiguous()
                gflow = torch.empty_like(ft) if flow.needs else None
                # the feature gradient is a scatter (atomicAdd): straight into x's accumulated gradient when there is one
                if x.needs:
                    def write(gx, acc):
                        if config.deterministic and C >= 16:
                            ws = torch.empty(int(E.call("cc_feature_warp_bwd_det_ws_bytes", B, C, H, W)), dtype=torch.uint8,
                                             device=gx.device)
                            E.call("cc_feature_warp_bwd_det", g, xt, ft, gflow, gx, ws, B, C, H, W, int(align_corners),
                                   float(flow_scale), int(acc), STREAM)
                            return
                        if not acc:
                            gx.zero_()
                        E.call("cc_feature_warp_bwd", g, xt, ft, gflow, gx, B, C, H, W, int(align_corners), float(flow_scale), STREAM)
                    x.from_writer(write)
                else:
                    x.skip()
                    E.call("cc_feature_warp_bwd", g, xt, ft, gflow, None, B, C, H, W, int(align_corners), float(flow_scale), STREAM)
                flow.add_alias(gflow) if flow.needs else flow.skip()
            self._node(bwd)
        return yt

    def crop(self, x, H, W):
        """x[:, :, :H, :W] (crop_like of the decoders; the identity at the sizes the networks are trained on)"""
        if x.t.shape[2] == H and x.t.shape[3] == W:
            return x
        y = x.t[:, :, :H, :W].contiguous()
        yt = self._tt(y, act=x.act, act_a=x.act_a, act_b=x.act_b)
        self._use(x)
        if self.record:
            def bwd():
                g, pre = yt.final()
                if g is None:
                    x.skip()
                    return
                if pre:                                       # cannot happen: a crop is consumed through a concatenation
                    raise RuntimeError("crop: gradient already passed through the activation")

                def write(gx, acc):
                    if not acc:
                        gx.zero_()
                    gx[:, :, :H, :W].add_(g)
                x.from_writer(write)
            yt.act = 0                                        # its consumers must not apply the producer's act'
            self._node(bwd)
        return yt

    # ------------------------------------------------------------------ small tails as stock torch (pose mean, occlusion softmax)
    def torch_fn(self, xs, fn):
        """[y, ...] = fn(*[x.t]) with stock torch ops (tiny tensors only); differentiated by torch.autograd inside the node.
        fn returns a tuple of tensors."""
        if not self.record:
            with torch.no_grad():
                return [self._tt(y) for y in fn(*[x.t for x in xs])]
        ins = [x.t.detach().requires_grad_(True) for x in xs]
        with torch.enable_grad():
            ys = list(fn(*ins))
        outs = [self._tt(y.detach()) for y in ys]
        self._use(*xs)

        def bwd():
            pairs = [(y, o.final()[0]) for y, o in zip(ys, outs)]
            pairs = [(y, g) for y, g in pairs if g is not None and y.requires_grad]
            if not pairs:
                for x in xs:
                    x.skip()
                return
            gs = torch.autograd.grad([y for y, _ in pairs], ins, [g for _, g in pairs], allow_unused=True)
            for x, g in zip(xs, gs):
                x.add_alias(g.contiguous()) if (g is not None and x.needs) else x.skip()
        self._node(bwd)
        return outs

    # ------------------------------------------------------------------ backward pass
    def backward(self, outputs, grads):
        """outputs: TTs returned by the network; grads: their upstream gradients (None = unused)."""
        assert self.record
        for tt in self.tts:
            tt.remaining = tt.uses
            tt.grad, tt.pend, tt.pre = None, [], False
        for o, g in zip(outputs, grads):
            if g is not None:
                o.pend.append(g if (g.dim() != 4 or _dense(g)) else g.contiguous())
        for fn in reversed(self.nodes):
            fn()


# ---------------------------------------------------------------------- one autograd node per network
class _NetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, body, n_in, *tensors):
        need = ctx.needs_input_grad[2:]
        tape = Tape(record=any(need))
        tape.owner = getattr(body, "__self__", None)
        ins = [tape.leaf(t, needs=need[i]) for i, t in enumerate(tensors[:n_in])]
        outs = body(tape, *ins)
        ctx.tape, ctx.ins, ctx.outs, ctx.n_in, ctx.body = tape, ins, outs, n_in, body
        ctx.params = tensors[n_in:]
        ctx.set_materialize_grads(False)
        return tuple(o.t for o in outs)

    @staticmethod
    def backward(ctx, *gouts):
        tape = ctx.tape
        tape.backward(ctx.outs, gouts)
        gin = [tt.final()[0] if tt.needs else None for tt in ctx.ins]
        gpar = []
        for p in ctx.params:
            ent = tape.param_grads.get(id(p))
            gpar.append(ent[1] if ent is not None else None)
        ctx.tape = None
        if NET_DONE is not None:
            NET_DONE(getattr(ctx.body, "__self__", None))
        return (None, None) + tuple(gin) + tuple(gpar)


def run_network(body, inputs, params):
    """body(tape, *input TTs) -> [output TTs], recorded and differentiated on a Tape; torch.autograd sees ONE node."""
    return _NetFn.apply(body, len(inputs), *inputs, *params)
