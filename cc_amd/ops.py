"""Network-layer operators of the engine: thin ``torch.autograd.Function`` bindings of the gfx950 kernels
(convolutions on fp32 MFMA, the 9x9 cost volume, feature warp) behind the C ABI.

Everything heavy is a HIP kernel.  What is still stock ATen (tiny tensors, listed in DESIGN.md as "torch
plumbing"): nearest upsampling + channel softmax of the (unused-by-training) occlusion head, torch.cat.
"""
import torch
import torch.nn.functional as F

from ._lib import engine, image_dense, STREAM
from .config import debug as _dbg

ACT = {None: 0, "relu": 1, "lrelu": 2, "sigmoid": 3}


def _c(t):
    return t.contiguous().float()


def _slice_or_c(t):
    """-> (tensor, batch stride in floats): a per-image dense channel slice (the narrow() views autograd hands to the producers
    of a torch.cat) is read in place through the kernels' batch-stride arguments instead of being copied."""
    if (t.dtype == torch.float32 and not t.is_contiguous() and image_dense(t) and t.data_ptr() % 16 == 0 and t.stride(0) % 4 == 0
            and not _dbg.no_slice_gy):
        return t, t.stride(0)
    t = _c(t)
    return t, t.shape[1] * t.shape[2] * t.shape[3]


def _ws(nbytes, ref):
    return torch.empty(max(int(nbytes) // 4, 4), device=ref.device, dtype=torch.float32)


# ----------------------------------------------------------------------------- per-step weight prepack
class PackRegistry:
    """The [tap][c][m] weight images of every conv layer (forward + data-gradient layouts), refreshed by cc_repack_table launches
    instead of one tiny repack launch inside each of the ~540 conv calls.  Two refresh protocols (cc_amd/trainer.py):
      * legacy step forms: ONE launch for all layers at the start of a trainer step (prepack_all), the images go stale when the
        optimizer runs behind the step (invalidate);
      * per-network pipeline (round 6): every network's images are rebuilt right behind ITS Adam segment, on that network's stream
        (repack_range), so the next step starts with fresh images and no launch (begin_step).
    Outside a trainer step `valid` is False and every conv call repacks for itself."""

    def __init__(self):
        self.entries = {}
        self.valid = False
        self.recording = False      # new layers are registered only inside a trainer step (not by eval passes at other sizes)
        self.dirty = False          # layers registered since the table was built
        self.table = None
        self.total_blocks = 0
        self.range_tables = {}      # (lo_ptr, hi_ptr) -> (n entries it was built from, table, blocks, entries)

    def get(self, kind, w, geom):
        """-> prepacked buffer (tensor) or None.  Unknown layers are registered for the next refresh."""
        key = (kind, w.data_ptr(), geom)
        ent = self.entries.get(key)
        if ent is None:
            if self.recording:
                self._register(key, kind, w, geom)
            return None
        if ent is False or not self.valid or not ent["ok"]:
            return None
        return ent["buf"]

    def _register(self, key, kind, w, geom):
        import ctypes
        E = engine()
        fn = "cc_conv2d_fwd_pack" if kind == "fwd" else "cc_conv2d_dgrad_pack"
        n = E.call(fn + "_floats", *geom)
        if n == 0:
            self.entries[key] = False
            return
        buf = torch.zeros(int(n), device=w.device, dtype=torch.float32)
        stride = int(geom[7])
        host = (ctypes.c_long * (16 * max(4, stride * stride)))()       # one 16-long descriptor per output parity class
        nd = E.fn[fn + "_desc"](*geom, w.data_ptr(), buf.data_ptr(), ctypes.addressof(host))
        descs = [[int(host[16 * i + k]) for k in range(16)] for i in range(nd)]
        self.entries[key] = {"buf": buf, "descs": descs, "ok": False, "w": w}
        self.dirty = True

    def ensure(self, kind, w, geom):
        """-> the image buffer of (kind, w, geom), registered on the spot when unknown (launch plans hold its address);
        its CONTENT is valid after the next refresh.  None: this geometry does not run on the patch kernel."""
        key = (kind, w.data_ptr(), geom)
        ent = self.entries.get(key)
        if ent is None:
            self._register(key, kind, w, geom)
            ent = self.entries[key]
        return ent["buf"] if ent else None

    @staticmethod
    def _table(live):
        rows, blk = [], 0
        for e in live:
            for d in e["descs"]:
                d = list(d)
                d[14] = blk
                blk += d[15]
                rows.append(d)
        return torch.tensor(rows, dtype=torch.int64, device=live[0]["buf"].device).contiguous(), blk

    def prepack_all(self):
        """Refresh every registered image with one launch (the weights changed) and open registration."""
        live = [e for e in self.entries.values() if e]
        self.recording = True
        if not live:
            return
        if self.dirty or self.table is None:
            self.table, self.total_blocks = self._table(live)
            self.dirty = False
        engine().call("cc_repack_table", self.table, self.table.shape[0], self.total_blocks, STREAM)
        for e in live:
            e["ok"] = True
        self.valid = True

    def begin_step(self):
        """Start of a trainer step of the per-network pipeline: the images were rebuilt behind the previous step's Adam segments --
        nothing to launch unless a layer is new or somebody marked the images stale (mark_stale)."""
        live = [e for e in self.entries.values() if e]
        if live and all(e["ok"] for e in live):
            self.recording = True
            self.valid = True
            return
        self.prepack_all()

    def repack_range(self, lo_ptr, hi_ptr):
        """Rebuild the images of the layers whose weights live in [lo_ptr, hi_ptr) (one network's segment of the parameter bucket,
        just updated by its Adam segment) with one launch on the current stream."""
        n_ent = len(self.entries)
        ent = self.range_tables.get((lo_ptr, hi_ptr))
        if ent is None or ent[0] != n_ent:
            live = [e for e in self.entries.values() if e and lo_ptr <= e["w"].data_ptr() < hi_ptr]
            tab, blk = self._table(live) if live else (None, 0)
            ent = self.range_tables[(lo_ptr, hi_ptr)] = (n_ent, tab, blk, live)
        _, tab, blk, live = ent
        if tab is None:
            return
        engine().call("cc_repack_table", tab, tab.shape[0], blk, STREAM)
        for e in live:
            e["ok"] = True

    def mark_stale(self):
        """the weights changed outside the trainer's own optimizer (load_state_dict, a user's in-place edit)"""
        for e in self.entries.values():
            if e:
                e["ok"] = False

    def end_step(self):
        """per-network pipeline: the images stay (they match the updated weights); conv calls outside a step repack for themselves"""
        self.valid = False
        self.recording = False

    def invalidate(self):
        """legacy step forms: the optimizer runs behind the step -- every image is stale afterwards"""
        self.valid = False
        self.recording = False
        self.mark_stale()

    def reset(self):
        """Forget every registered layer (tests; a new set of networks)."""
        self.entries.clear()
        self.range_tables.clear()
        self.valid, self.dirty, self.table, self.total_blocks = False, False, None, 0


packs = PackRegistry()

# parameter data_ptr -> gradient buffer (a view of the optimizer's flat gradient bucket).  When a conv weight / bias is
# registered here (trainer.FlatAdam does it), its gradient is ACCUMULATED into that buffer by the wgrad / bias kernels
# themselves and autograd receives None: no per-parameter `grad += new` launch (~330 per step), same arithmetic.
grad_sinks = {}


def _sink(p):
    return grad_sinks.get(p.data_ptr()) if grad_sinks else None


class _WgradQueue:
    """Weight gradients of same-shaped layers (the four 3x3 convolutions of a DispResNet6 stage, ...) are independent of the
    rest of the backward pass once their operands exist: when the trainer enables the queue, a layer's weight-gradient call
    is parked (operands kept alive) and launched together with its shape-mates as ONE cc_conv2d_wgrad_group launch (+ one
    reduction) -- more workgroups per launch, fewer partial slabs -- when 4 have gathered or at the end of the backward stage.
    Only for gradients that are accumulated into the optimizer's flat bucket (nothing is returned to autograd)."""
    GROUP = 4

    def __init__(self):
        self.enabled = False
        self.pending = {}

    def push(self, key, a, x, gw):
        q = self.pending.setdefault(key, [])
        if any(t[2].data_ptr() == gw.data_ptr() for t in q):      # a weight used twice: two `gw +=` of ONE launch would race
            self._launch(key)
            q = self.pending.setdefault(key, [])
        q.append((a, x, gw))
        if len(q) >= self.GROUP:
            self._launch(key)

    def _take_items(self):
        items = []
        for key in list(self.pending):
            q = self.pending.pop(key, None)
            if not q:
                continue
            B, M, AH, AW, Cin, IH, IW, R, S, si, pad, o_sm, o_sc, a_bs, x_bs = key
            for g0 in range(0, len(q), self.GROUP):
                qq = q[g0:g0 + self.GROUP]
                items.append(([t[0] for t in qq], [t[1] for t in qq], [t[2] for t in qq],
                              (B, M, AH, AW, a_bs, Cin, IH, IW, x_bs, R, S, si, pad, o_sm, o_sc)))
        return items

    def _launch(self, key):
        q = self.pending.pop(key, None)
        if not q:
            return
        B, M, AH, AW, Cin, IH, IW, R, S, si, pad, o_sm, o_sc, a_bs, x_bs = key
        _wgrad_group([t[0] for t in q], [t[1] for t in q], [t[2] for t in q], q[0][0], B, M, AH, AW, a_bs, Cin, IH, IW,
                     x_bs, R, S, si, pad, o_sm, o_sc, 1)

    def flush(self):
        # what is still parked at the end of the stage (single layers of unique shapes, incomplete groups) goes out as ONE list call:
        # the groups that take the generic weight-gradient kernel share launches (cc_conv2d_wgrad_list)
        items = self._take_items()
        if items:
            _wgrad_list(items)
        wgrad_reduces._cur().flush()          # (this stream's reductions: the proxy runs every stream's queue on its own stream)

    def busy(self):
        return any(self.pending.values())

    def drop(self):
        """error path: forget the parked launches (their operands die with the failed step)"""
        self.pending = {}


class _WgradReduces:
    """Second stages (sums of the split-K partial slabs) of the weight-gradient launches of a backward stage, parked while the
    trainer's queue is enabled and run as ONE cc_wgrad_reduce_table launch per 32 at the end of the stage: ~130 reductions of
    8-15 us each per step become 5 launches that fill the chip.  The workspaces stay alive until then."""

    def __init__(self):
        self.desc = []
        self.keep = []
        self.targets = set()
        self.bias_jobs = []      # parked pure bias-gradient sums (cc_bias_grad_defer): one cc_bias_grad_table launch per 32

    def flush(self):
        import ctypes
        if self.bias_jobs:
            arr = (ctypes.c_long * len(self.bias_jobs))(*self.bias_jobs)
            engine().call("cc_bias_grad_table", ctypes.addressof(arr), len(self.bias_jobs) // 12, STREAM)
        if self.desc:
            if _dbg.reduce_trace is not None:       # tools/reduce_bytes.py: who writes how many partial-slab bytes
                _dbg.reduce_trace.extend(self.desc)
            arr = (ctypes.c_long * len(self.desc))(*self.desc)
            engine().call("cc_wgrad_reduce_table", ctypes.addressof(arr), len(self.desc) // 16, STREAM)
        self.desc, self.keep, self.targets, self.bias_jobs = [], [], set(), []

    def busy(self):
        return bool(self.desc or self.bias_jobs)

    def drop(self):
        self.desc, self.keep, self.targets, self.bias_jobs = [], [], set(), []

    def park_bias(self, gy, gbias, B, C, H, W, gy_bs):
        """gbias += sum over (n, h, w) of gy, at the end of the stage (gy is kept alive until then)"""
        import ctypes
        if gbias.data_ptr() in self.targets:
            self.flush()
        self.targets.add(gbias.data_ptr())
        E = engine()
        ws = _ws(E.call("cc_act_bwd_ws_bytes", C), gy)
        job, red, nred = (ctypes.c_long * 12)(), (ctypes.c_long * 16)(), ctypes.c_int(0)
        E.call("cc_bias_grad_defer", gy, gbias, ws, B, C, H, W, gy_bs, 1, ctypes.addressof(job), ctypes.addressof(red),
               ctypes.addressof(nred))
        self.bias_jobs.extend(job[:])
        if nred.value:
            self.desc.extend(red[:])
        self.keep.append((ws, gbias, gy))


def _stream_key():
    """(key, torch stream or None) of the stream the caller launches on.  The parked work of a backward stage is kept PER STREAM:
    with config.net_streams the networks run on streams of their own, and a shape-mate group / reduce table launched on one stream
    must not read operands another stream is still producing."""
    if torch.cuda.is_available():
        st = torch.cuda.current_stream()
        return st.cuda_stream, st
    return 0, None


class _PerStream:
    """Proxy of one parked-work object per stream.  Attribute access goes to the current stream's instance; flush() runs every
    instance's flush ON ITS OWN STREAM (ordered behind the kernels that produced its operands); the caller joins the streams."""

    def __init__(self, factory):
        object.__setattr__(self, "_factory", factory)
        object.__setattr__(self, "_inst", {})
        object.__setattr__(self, "enabled", False)

    def _cur(self):
        key, st = _stream_key()
        ent = self._inst.get(key)
        if ent is None:
            ent = self._inst[key] = (self._factory(), st)
        return ent[0]

    def __getattr__(self, name):
        return getattr(self._cur(), name)

    def __setattr__(self, name, value):
        if name == "enabled":
            object.__setattr__(self, name, value)
        else:
            setattr(self._cur(), name, value)

    def flush(self):
        """every stream's parked work, launched on ITS stream -> the streams touched (the caller joins them)"""
        touched = []
        for key in list(self._inst):
            inst, st = self._inst[key]
            if st is None or st.cuda_stream == torch.cuda.current_stream().cuda_stream:
                inst.flush()
            else:
                touched.append(st)
                with torch.cuda.stream(st):
                    inst.flush()
        self._inst.clear()          # (all empty now; streams() lists what has parked work since the last flush)
        return touched

    def drop(self):
        """error path of a trainer step: forget every stream's parked work WITHOUT launching it (no fork / join is in place, and the
        operands belong to a step that failed)"""
        for inst, _ in self._inst.values():
            inst.drop()
        self._inst.clear()

    def streams(self):
        """the streams that HAVE parked work: the caller forks them before flush() and joins them after it.  (Streams whose queue a
        network's own tail has emptied already are left alone: an empty fork / join pair is still a cross-stream edge of the
        captured graph.)"""
        return [st for inst, st in self._inst.values() if st is not None and inst.busy()]


wgrad_queue = _PerStream(_WgradQueue)
wgrad_reduces = _PerStream(_WgradReduces)


def _act_bwd_bias(gys, ys, geffs, gbs, ref, B, C, H, W, gy_bs, act, act_a, act_b, accumulate):
    """cc_act_bwd_bias_group for G = len(gys) <= 4 same-shaped problems (lists may hold None uniformly).  Inside a trainer stage
    the second stage of bias gradients that accumulate into the optimizer's bucket is parked with the weight-gradient
    reductions (one table launch per stage instead of one k_bias_reduce per call)."""
    import ctypes
    E = engine()
    G = len(gys)
    a1, a2, a3, a4 = _parr(gys), _parr(ys), _parr(geffs), _parr(gbs)
    has_y, has_ge, has_gb = ys[0] is not None, geffs[0] is not None, gbs[0] is not None
    ws = _ws(E.call("cc_act_bwd_ws_bytes", C) * G, ref)
    args = (G, _addr(a1), _addr(a2) if has_y else 0, _addr(a3) if has_ge else 0, _addr(a4) if has_gb else 0, ws, B, C, H, W,
            gy_bs, C * H * W, C * H * W, act, act_a, act_b, int(accumulate))
    if has_gb and accumulate and wgrad_queue.enabled and not _dbg.no_wgrad_defer:
        ptrs = [t.data_ptr() for t in gbs]
        if wgrad_reduces.targets.intersection(ptrs):
            wgrad_reduces._cur().flush()       # (this stream's table only: the proxy's flush() would launch every stream's)
        wgrad_reduces.targets.update(ptrs)
        red = (ctypes.c_long * (16 * G))()
        nred = ctypes.c_int(0)
        E.call("cc_act_bwd_bias_group_defer", *args, ctypes.addressof(red), G, ctypes.addressof(nred), STREAM)
        if nred.value:
            wgrad_reduces.keep.append((ws, gbs))
            wgrad_reduces.desc.extend(red[:16 * nred.value])
    else:
        E.call("cc_act_bwd_bias_group", *args, STREAM)


_WS_BYTES = {}


def _wgrad_ws_bytes(B, M, AH, AW, Cin, R, S, si):
    """cc_conv2d_wgrad_ws_bytes, memoised per geometry (the library plans every group size for it: once per shape, not once per
    weight-gradient call of an eager step)"""
    E = engine()
    tools = getattr(E, "_is_tools", None)
    if tools is None:
        tools = E._is_tools = bool(E.fn["cc_is_tools_build"]())
    if tools:                     # (tools / emulation builds: the plan depends on the environment switches of the moment)
        return E.call("cc_conv2d_wgrad_ws_bytes", B, M, AH, AW, Cin, R, S, si)
    key = (id(E), B, M, AH, AW, Cin, R, S, si)
    v = _WS_BYTES.get(key)
    if v is None:
        v = _WS_BYTES[key] = E.call("cc_conv2d_wgrad_ws_bytes", B, M, AH, AW, Cin, R, S, si)
    return v


_ZEROS64 = {}


def _zeros64(ref):
    """64 zero floats per device, never written: the halo source of the LDS-DMA weight-gradient kernels"""
    z = _ZEROS64.get(ref.device)
    if z is None:
        z = _ZEROS64[ref.device] = torch.zeros(64, device=ref.device, dtype=torch.float32)
    return z


def _wgrad_group(a_list, x_list, gw_list, ref, B, M, AH, AW, a_bs, Cin, IH, IW, x_bs, R, S, si, pad, o_sm, o_sc, accumulate):
    """cc_conv2d_wgrad_group for G = len(a_list) <= 4 same-shaped problems.  Inside a trainer stage (queue enabled) and when the
    results go straight into the optimizer's gradient bucket (accumulate), the reductions are parked (see _WgradReduces)."""
    import ctypes
    E = engine()
    G = len(a_list)
    per = _wgrad_ws_bytes(B, M, AH, AW, Cin, R, S, si)
    a1, a2, a3 = _parr(a_list), _parr(x_list), _parr(gw_list)
    if accumulate and wgrad_queue.enabled and not _dbg.no_wgrad_defer:
        ptrs = [t.data_ptr() for t in gw_list]
        if len(set(ptrs)) != len(ptrs):                    # one weight twice in ONE launch: its two `gw +=` would race -> one by one
            for a, x, gw in zip(a_list, x_list, gw_list):
                _wgrad_group([a], [x], [gw], ref, B, M, AH, AW, a_bs, Cin, IH, IW, x_bs, R, S, si, pad, o_sm, o_sc, accumulate)
            return
        if wgrad_reduces.targets.intersection(ptrs):      # a weight used twice in one stage: two parked `gw +=` would race
            wgrad_reduces._cur().flush()
        wgrad_reduces.targets.update(ptrs)
        red = (ctypes.c_long * (16 * G))()
        nred = ctypes.c_int(0)
        ws = _ws(per * G, ref)
        E.call("cc_conv2d_wgrad_group_defer", G, _addr(a1), _addr(a2), _addr(a3), ws, B, M, AH, AW, a_bs, Cin, IH, IW, x_bs, R, S, si,
               pad, o_sm, o_sc, 1, _zeros64(ref), ctypes.addressof(red), G, ctypes.addressof(nred), STREAM)
        if nred.value:
            wgrad_reduces.keep.append((ws, gw_list))
            wgrad_reduces.desc.extend(red[:16 * nred.value])
    else:
        ws = _ws(per * G, ref)
        E.call("cc_conv2d_wgrad_group", G, _addr(a1), _addr(a2), _addr(a3), ws, B, M, AH, AW, a_bs, Cin, IH, IW, x_bs, R, S, si, pad,
               o_sm, o_sc, int(accumulate), STREAM)


def _wgrad_list(items):
    """items: [(a_list, x_list, gw_list, (B, M, AH, AW, a_bs, Cin, IH, IW, x_bs, R, S, si, pad, o_sm, o_sc))], gradients accumulated into
    the optimizer's bucket, reductions parked (the queue is enabled): one cc_conv2d_wgrad_list call."""
    import ctypes
    E = engine()
    if _dbg.no_wgrad_defer or not wgrad_queue.enabled or _dbg.no_wgrad_list:
        for a_list, x_list, gw_list, geo in items:
            B, M, AH, AW, a_bs, Cin, IH, IW, x_bs, R, S, si, pad, o_sm, o_sc = geo
            _wgrad_group(a_list, x_list, gw_list, a_list[0], B, M, AH, AW, a_bs, Cin, IH, IW, x_bs, R, S, si, pad, o_sm, o_sc, 1)
        return
    desc, keep, cap = [], [], 0

    def run():
        nonlocal desc, keep, cap
        if not desc:
            return
        arr = (ctypes.c_long * len(desc))(*desc)
        red = (ctypes.c_long * (16 * cap))()
        nred = ctypes.c_int(0)
        E.call("cc_conv2d_wgrad_list", len(desc) // 32, ctypes.addressof(arr), _zeros64(keep[0][0]), ctypes.addressof(red), cap,
               ctypes.addressof(nred), STREAM)
        if nred.value:
            wgrad_reduces.desc.extend(red[:16 * nred.value])
        wgrad_reduces.keep.extend(keep)
        desc, keep, cap = [], [], 0

    for a_list, x_list, gw_list, geo in items:
        B, M, AH, AW, a_bs, Cin, IH, IW, x_bs, R, S, si, pad, o_sm, o_sc = geo
        ptrs = [t.data_ptr() for t in gw_list]
        if len(set(ptrs)) != len(ptrs) or wgrad_reduces.targets.intersection(ptrs):
            run()               # (their slabs must exist before the flush the per-group path may trigger)
            _wgrad_group(a_list, x_list, gw_list, a_list[0], B, M, AH, AW, a_bs, Cin, IH, IW, x_bs, R, S, si, pad, o_sm, o_sc, 1)
            continue
        wgrad_reduces.targets.update(ptrs)
        G = len(a_list)
        ws = _ws(_wgrad_ws_bytes(B, M, AH, AW, Cin, R, S, si) * G, a_list[0])
        pad4 = lambda ts: [t.data_ptr() for t in ts] + [0] * (4 - len(ts))
        desc.extend([G] + pad4(a_list) + pad4(x_list) + pad4(gw_list) +
                    [ws.data_ptr(), B, M, AH, AW, a_bs, Cin, IH, IW, x_bs, R, S, si, pad, o_sm, o_sc, 1, 0, 0])
        keep.append((ws, gw_list, a_list, x_list))
        cap += G
    run()


# ----------------------------------------------------------------------------- convolution
class _Conv2dFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, bias, res, stride, pad, act, act_a, act_b, pre_act=0, pre_b=0.0, defer=False):
        x, w = _c(x), _c(w)
        B, Cin, IH, IW = x.shape
        Cout, _, R, S = w.shape
        OH = (IH + 2 * pad - R) // stride + 1
        OW = (IW + 2 * pad - S) // stride + 1
        y = torch.empty(B, Cout, OH, OW, device=x.device, dtype=torch.float32)
        bias_c = None if bias is None else _c(bias)
        res_c = None if res is None else _c(res)
        E = engine()
        ws = _ws(E.call("cc_conv2d_fwd_ws_bytes", B, Cin, IH, IW, Cout, R, S, stride, pad, OH, OW), x)
        pk = packs.get("fwd", w, (B, Cin, IH, IW, Cout, R, S, stride, pad, OH, OW))
        E.call("cc_conv2d_fwd", x, w, bias_c, res_c, y, ws, pk, B, Cin, IH, IW, Cin * IH * IW, Cout, R, S, stride, pad,
               OH, OW, Cout * OH * OW, Cout * OH * OW, act, act_a, act_b, STREAM)
        # defer: the consumers of y (convolutions called with pre_act = this activation) apply act'(y) in their data-gradient
        # epilogue, so the gradient arriving here is already w.r.t. the pre-activation (see _ConvGroupFn)
        ctx.save_for_backward(x, w, y if (act != 0 and not defer) else None)
        ctx.cfg = (stride, pad, act if not defer else 0, act_a, act_b, bias is not None, res is not None)
        ctx.pre = (pre_act, pre_b)
        ctx.bias_ptr = bias_c.data_ptr() if bias_c is not None else 0
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w, y = ctx.saved_tensors
        stride, pad, act, act_a, act_b, has_bias, has_res = ctx.cfg
        pre_act, pre_b = ctx.pre
        E = engine()
        # a layer with an activation passes gy through cc_act_bwd_bias first: that call reads a concat-gradient slice in place
        gy, gy_bs = _slice_or_c(gy) if act != 0 else (_c(gy), 0)
        B, Cin, IH, IW = x.shape
        Cout, _, R, S = w.shape
        OH, OW = gy.shape[2], gy.shape[3]
        need = ctx.needs_input_grad
        gbias, bsink = None, None
        if has_bias and need[2]:
            bsink = grad_sinks.get(ctx.bias_ptr) if grad_sinks else None
            gbias = bsink if bsink is not None else torch.empty(Cout, device=x.device, dtype=torch.float32)
        if act != 0 or gbias is not None:
            geff = torch.empty(gy.shape, device=gy.device, dtype=torch.float32) if act != 0 else None
            _act_bwd_bias([gy], [y if act != 0 else None], [geff], [gbias], x, B, Cout, OH, OW,
                          gy_bs if act != 0 else Cout * OH * OW, act, act_a, act_b, bsink is not None)
            if geff is not None:
                gy = geff
        if bsink is not None:
            gbias = None
        gx = gw = None
        if need[0]:
            gx = torch.empty_like(x)
            ws = _ws(E.call("cc_conv2d_dgrad_ws_bytes", B, Cout, OH, OW, Cin, R, S, stride, pad, IH, IW), x)
            pk = packs.get("dgrad", w, (B, Cout, OH, OW, Cin, R, S, stride, pad, IH, IW, Cin * R * S, R * S))
            if pre_act:       # gradient w.r.t. the producer's pre-activation: act'(x) multiplied in the epilogue
                a1, a2, a3, a4, a5 = _parr([gy]), _parr([w]), _parr([gx]), _parr([x]), _parr([pk])
                E.call("cc_conv2d_dgrad_group", 1, _addr(a1), _addr(a2), 0, _addr(a3), _addr(a4), ws, _addr(a5), B, Cout, OH, OW,
                       Cout * OH * OW, Cin, R, S, stride, pad, IH, IW, Cin * IH * IW, Cin * IH * IW, Cin * R * S, R * S, pre_act, 1.0,
                       pre_b, STREAM)
            else:
                E.call("cc_conv2d_dgrad", gy, w, None, gx, ws, pk, B, Cout, OH, OW, Cout * OH * OW, Cin, R, S, stride, pad, IH, IW,
                       Cin * IH * IW, Cin * R * S, R * S, 0, 1.0, 0.0, STREAM)
        if need[1]:
            wsink = _sink(w)
            if wsink is not None and wgrad_queue.enabled:
                wgrad_queue.push((B, Cout, OH, OW, Cin, IH, IW, R, S, stride, pad, Cin * R * S, R * S, Cout * OH * OW, Cin * IH * IW),
                                 gy, x, wsink)
            else:
                gw = wsink if wsink is not None else torch.empty_like(w)
                _wgrad_group([gy], [x], [gw], x, B, Cout, OH, OW, Cout * OH * OW, Cin, IH, IW, Cin * IH * IW, R, S, stride, pad,
                             Cin * R * S, R * S, int(wsink is not None))
                if wsink is not None:
                    gw = None
        gres = gy if (has_res and need[3]) else None
        return gx, gw, gbias, gres, None, None, None, None, None, None, None, None


# ----------------------------------------------------------------------------- grouped convolution (parallel branches)
def _parr(ts):
    """HOST array of device addresses (0 = null) for the *_group entry points; keep the returned object alive over the call."""
    import ctypes
    return (ctypes.c_long * len(ts))(*[(t.data_ptr() if t is not None else 0) for t in ts])


def _addr(arr):
    import ctypes
    return ctypes.addressof(arr)


class _ConvGroupFn(torch.autograd.Function):
    """G same-shaped convolutions (with their own inputs, weights, biases) as ONE launch per pass (cc_conv2d_*_group).

    meta = (G, stride, pad, act, act_a, act_b, has_bias, pre_act, pre_b, defer)
      defer   : this layer's outputs feed ONLY convolutions called with pre_act = this activation; those multiply act'(y)
                into their data-gradient epilogue, so the gradient arriving here is already w.r.t. the pre-activation
                (no separate activation-backward pass over gy and y);
      pre_act : activation (code, slope pre_b) of the deferring layer that produced xs.
    tensors = xs + weights (+ biases)."""

    @staticmethod
    def forward(ctx, meta, *tensors):
        G, stride, pad, act, act_a, act_b, has_bias, pre_act, pre_b, defer = meta
        xs = [_c(t) for t in tensors[:G]]
        ws = [_c(t) for t in tensors[G:2 * G]]
        bs = [_c(t) for t in tensors[2 * G:3 * G]] if has_bias else [None] * G
        B, Cin, IH, IW = xs[0].shape
        Cout, _, R, S = ws[0].shape
        OH = (IH + 2 * pad - R) // stride + 1
        OW = (IW + 2 * pad - S) // stride + 1
        E = engine()
        ys = [torch.empty(B, Cout, OH, OW, device=xs[0].device, dtype=torch.float32) for _ in range(G)]
        geom = (B, Cin, IH, IW, Cout, R, S, stride, pad, OH, OW)
        wsb = _ws(E.call("cc_conv2d_fwd_group_ws_bytes", G, *geom), xs[0])
        pks = [packs.get("fwd", w, geom) for w in ws]
        ax, aw, ab, ay, ap = _parr(xs), _parr(ws), _parr(bs), _parr(ys), _parr(pks)
        E.call("cc_conv2d_fwd_group", G, _addr(ax), _addr(aw), _addr(ab), 0, _addr(ay), wsb, _addr(ap), B, Cin, IH, IW,
               Cin * IH * IW, Cout, R, S, stride, pad, OH, OW, Cout * OH * OW, 0, act, act_a, act_b, STREAM)
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(*xs, *ws, *([y for y in ys] if (act != 0 and not defer) else []))
        ctx.meta = meta
        ctx.bias_ptrs = [b.data_ptr() if b is not None else 0 for b in bs]
        return tuple(ys)

    @staticmethod
    def backward(ctx, *gys):
        G, stride, pad, act, act_a, act_b, has_bias, pre_act, pre_b, defer = ctx.meta
        saved = ctx.saved_tensors
        xs, ws = saved[:G], saved[G:2 * G]
        ys = saved[2 * G:] if len(saved) > 2 * G else [None] * G
        need = ctx.needs_input_grad
        E = engine()
        B, Cin, IH, IW = xs[0].shape
        Cout, _, R, S = ws[0].shape
        live = [k for k in range(G) if gys[k] is not None]          # branches nobody differentiates (occlusion decoders) drop out
        gx_out, gw_out, gb_out = [None] * G, [None] * G, [None] * G
        if not live:
            return (None,) + tuple(gx_out + gw_out + (gb_out if has_bias else []))
        gy = {k: _c(gys[k]) for k in live}
        OH, OW = gy[live[0]].shape[2], gy[live[0]].shape[3]
        n = len(live)
        dev = xs[0].device
        # ---- activation backward (unless deferred to the consumers' data-gradient epilogues) + bias gradients
        want_b = [k for k in live if has_bias and need[1 + 2 * G + k]]
        bsinks = {k: (grad_sinks.get(ctx.bias_ptrs[k]) if grad_sinks else None) for k in want_b}
        all_sink = bool(want_b) and all(v is not None for v in bsinks.values())
        do_act = act != 0 and not defer
        if do_act or want_b:
            idx = live if do_act else want_b
            geff = {k: torch.empty_like(gy[k]) for k in idx} if do_act else {}
            gb = {}
            for k in idx:
                if k in want_b:
                    gb[k] = bsinks[k] if all_sink else torch.empty(Cout, device=dev, dtype=torch.float32)
            # bias gradients are wanted for all of idx or for none of it (one network: all parameters trainable or all frozen)
            with_b = len(gb) == len(idx)
            for c0 in range(0, len(idx), 4):
                ch = idx[c0:c0 + 4]
                _act_bwd_bias([gy[k] for k in ch], [ys[k] if do_act else None for k in ch], [geff.get(k) for k in ch],
                              [gb.get(k) if with_b else None for k in ch], xs[0], B, Cout, OH, OW, Cout * OH * OW,
                              act if do_act else 0, act_a, act_b, all_sink)
            if not with_b and gb:          # mixed case: the remaining bias gradients one by one
                for k in gb:
                    E.call("cc_act_bwd_bias", geff.get(k, gy[k]), None, None, gb[k], _ws(E.call("cc_act_bwd_ws_bytes", Cout), xs[0]),
                           B, Cout, OH, OW, Cout * OH * OW, 0, 0, 0, 1.0, 0.0, int(all_sink), STREAM)
            if do_act:
                gy = geff
            if not all_sink:
                for k in gb:
                    gb_out[k] = gb[k]
        # ---- data gradients
        dx = [k for k in live if need[1 + k]]
        if dx:
            geom = (B, Cout, OH, OW, Cin, R, S, stride, pad, IH, IW)
            for k in dx:
                gx_out[k] = torch.empty_like(xs[k])
            wsb = _ws(E.call("cc_conv2d_dgrad_group_ws_bytes", len(dx), *geom), xs[0])
            pks = [packs.get("dgrad", ws[k], geom + (Cin * R * S, R * S)) for k in dx]
            a1, a2, a3 = _parr([gy[k] for k in dx]), _parr([ws[k] for k in dx]), _parr([gx_out[k] for k in dx])
            a4, a5 = _parr([xs[k] if pre_act else None for k in dx]), _parr(pks)
            E.call("cc_conv2d_dgrad_group", len(dx), _addr(a1), _addr(a2), 0, _addr(a3), _addr(a4) if pre_act else 0, wsb, _addr(a5),
                   B, Cout, OH, OW, Cout * OH * OW, Cin, R, S, stride, pad, IH, IW, Cin * IH * IW, Cin * IH * IW, Cin * R * S, R * S,
                   pre_act, 1.0, pre_b, STREAM)
        # ---- weight gradients
        dw = [k for k in live if need[1 + G + k]]
        if dw:
            wsinks = {k: _sink(ws[k]) for k in dw}
            all_wsink = all(v is not None for v in wsinks.values())
            gw = {k: (wsinks[k] if all_wsink else torch.empty_like(ws[k])) for k in dw}
            for c0 in range(0, len(dw), 4):
                ch = dw[c0:c0 + 4]
                _wgrad_group([gy[k] for k in ch], [xs[k] for k in ch], [gw[k] for k in ch], xs[0], B, Cout, OH, OW, Cout * OH * OW,
                             Cin, IH, IW, Cin * IH * IW, R, S, stride, pad, Cin * R * S, R * S, int(all_wsink))
            if not all_wsink:
                for k in dw:
                    gw_out[k] = gw[k]
        return (None,) + tuple(gx_out + gw_out + (gb_out if has_bias else []))


def conv2d_group(xs, weights, biases=None, stride=1, padding=0, act=None, slope=0.0, pre_act=None, pre_slope=0.0, defer=False):
    """[act(conv2d(x_k, w_k, b_k)) for k] for same-shaped problems, one launch per pass.  act='lrelu': slope (0 -> 0.2).
    See _ConvGroupFn for pre_act / defer (activation backward fused into the consumer's data-gradient epilogue)."""
    G = len(xs)
    has_bias = biases is not None and biases[0] is not None
    meta = (G, int(stride), int(padding), ACT[act], 1.0, float(slope), has_bias, ACT[pre_act], float(pre_slope), bool(defer))
    out = _ConvGroupFn.apply(meta, *xs, *weights, *(biases if has_bias else ()))
    return list(out)


def conv2d(x, w, bias=None, stride=1, padding=0, act=None, residual=None, act_a=1.0, act_b=0.0, pre_act=None, pre_slope=0.0,
           defer=False):
    """act(conv2d(x, w, bias) + residual): nn.Conv2d (+ fused ReLU / LeakyReLU / a*sigmoid+b epilogue).
    act='lrelu': act_b is the negative slope (0 -> the 0.2 of Back2Future).
    defer / pre_act: activation backward of a layer whose output feeds ONLY convolutions is applied by those consumers'
    data-gradient epilogue (producer: defer=True; every consumer: pre_act=<the producer's activation>)."""
    return _Conv2dFn.apply(x, w, bias, residual, int(stride), int(padding), ACT[act], float(act_a), float(act_b),
                           ACT[pre_act], float(pre_slope), bool(defer))


class _ConvT2dFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, bias, stride, pad, out_pad, act, act_b=0.0):
        x, w = _c(x), _c(w)
        B, Cin, IH, IW = x.shape
        _, Cout, R, S = w.shape
        OH = (IH - 1) * stride - 2 * pad + R + out_pad
        OW = (IW - 1) * stride - 2 * pad + S + out_pad
        y = torch.empty(B, Cout, OH, OW, device=x.device, dtype=torch.float32)
        bias_c = None if bias is None else _c(bias)
        # ConvTranspose2d forward == the transposed-conv arithmetic of cc_conv2d_dgrad with K = Cin, C = Cout
        E = engine()
        ws = _ws(E.call("cc_conv2d_dgrad_ws_bytes", B, Cin, IH, IW, Cout, R, S, stride, pad, OH, OW), x)
        pk = packs.get("dgrad", w, (B, Cin, IH, IW, Cout, R, S, stride, pad, OH, OW, Cout * R * S, R * S))
        E.call("cc_conv2d_dgrad", x, w, bias_c, y, ws, pk, B, Cin, IH, IW, Cin * IH * IW, Cout, R, S, stride, pad, OH, OW,
               Cout * OH * OW, Cout * R * S, R * S, act, 1.0, float(act_b), STREAM)
        ctx.save_for_backward(x, w, y if act != 0 else None)
        ctx.cfg = (stride, pad, act, bias is not None)
        ctx.act_b = float(act_b)
        ctx.bias_ptr = bias_c.data_ptr() if bias_c is not None else 0
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w, y = ctx.saved_tensors
        stride, pad, act, has_bias = ctx.cfg
        E = engine()
        gy, gy_bs = _slice_or_c(gy) if act != 0 else (_c(gy), 0)       # see _Conv2dFn.backward
        B, Cin, IH, IW = x.shape
        _, Cout, R, S = w.shape
        OH, OW = gy.shape[2], gy.shape[3]
        need = ctx.needs_input_grad
        gbias, bsink = None, None
        if has_bias and need[2]:
            bsink = grad_sinks.get(ctx.bias_ptr) if grad_sinks else None
            gbias = bsink if bsink is not None else torch.empty(Cout, device=x.device, dtype=torch.float32)
        if act != 0 or gbias is not None:
            geff = torch.empty(gy.shape, device=gy.device, dtype=torch.float32) if act != 0 else None
            _act_bwd_bias([gy], [y if act != 0 else None], [geff], [gbias], x, B, Cout, OH, OW,
                          gy_bs if act != 0 else Cout * OH * OW, act, 1.0, ctx.act_b, bsink is not None)
            if geff is not None:
                gy = geff
        if bsink is not None:
            gbias = None
        gx = gw = None
        if need[0]:
            # d/dx of a transposed conv is a plain strided conv of gy; the [Cin,Cout,R,S] weight IS its [M,C,R,S] weight
            gx = torch.empty_like(x)
            ws = _ws(E.call("cc_conv2d_fwd_ws_bytes", B, Cout, OH, OW, Cin, R, S, stride, pad, IH, IW), x)
            pk = packs.get("fwd", w, (B, Cout, OH, OW, Cin, R, S, stride, pad, IH, IW))
            E.call("cc_conv2d_fwd", gy, w, None, None, gx, ws, pk, B, Cout, OH, OW, Cout * OH * OW, Cin, R, S, stride, pad, IH, IW,
                   Cin * IH * IW, 0, 0, 1.0, 0.0, STREAM)
        if need[1]:
            wsink = _sink(w)
            gw = wsink if wsink is not None else torch.empty_like(w)
            _wgrad_group([x], [gy], [gw], x, B, Cin, IH, IW, Cin * IH * IW, Cout, OH, OW, Cout * OH * OW, R, S, stride, pad,
                         Cout * R * S, R * S, int(wsink is not None))
            if wsink is not None:
                gw = None
        return gx, gw, gbias, None, None, None, None, None


def conv_transpose2d(x, w, bias=None, stride=1, padding=0, output_padding=0, act=None, act_b=0.0):
    """act='lrelu': act_b is the negative slope (0 -> 0.2)."""
    return _ConvT2dFn.apply(x, w, bias, int(stride), int(padding), int(output_padding), ACT[act], float(act_b))


# ----------------------------------------------------------------------------- batch norm / upsampling
class _BNTrainFn(torch.autograd.Function):
    """nn.BatchNorm2d in training mode on the HIP kernels (csrc/bnorm.hip)."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, momentum, eps):
        x = _c(x)
        B, C, H, W = x.shape
        E = engine()
        y = torch.empty_like(x)
        mean = torch.empty(C, device=x.device, dtype=torch.float32)
        invstd = torch.empty_like(mean)
        E.call("cc_bn_train_fwd", x, weight, bias, running_mean, running_var, y, mean, invstd,
               _ws(E.call("cc_bn_ws_bytes", C), x), B, C, H, W, float(momentum), float(eps), STREAM)
        ctx.save_for_backward(x, weight, mean, invstd)
        ctx.ptrs = (weight.data_ptr() if weight is not None else 0, bias.data_ptr() if bias is not None else 0,
                    bias is not None)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight, mean, invstd = ctx.saved_tensors
        wptr, bptr, has_bias = ctx.ptrs
        gy = _c(gy)
        B, C, H, W = x.shape
        E = engine()
        need = ctx.needs_input_grad
        gx = torch.empty_like(x)
        wsink = grad_sinks.get(wptr) if (grad_sinks and weight is not None and need[1]) else None
        bsink = grad_sinks.get(bptr) if (grad_sinks and has_bias and need[2]) else None
        sink = wsink is not None and (bsink is not None or not (has_bias and need[2]))
        if sink:
            gw, gb = wsink, bsink
        else:
            gw = torch.empty(C, device=x.device, dtype=torch.float32) if (weight is not None and need[1]) else None
            gb = torch.empty(C, device=x.device, dtype=torch.float32) if (has_bias and need[2]) else None
        E.call("cc_bn_train_bwd", gy, x, weight, mean, invstd, gx, gw, gb, _ws(E.call("cc_bn_ws_bytes", C), x), B, C, H, W,
               int(sink), STREAM)
        if sink:
            gw = gb = None
        return gx, gw, gb, None, None, None, None


class _BNEvalFn(torch.autograd.Function):
    """nn.BatchNorm2d in eval mode (running statistics) on csrc/bnorm.hip; differentiable w.r.t. its input."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, eps):
        x = _c(x)
        B, C, H, W = x.shape
        y = torch.empty_like(x)
        engine().call("cc_bn_eval_fwd", x, weight, bias, running_mean, running_var, y, _ws(8 * C, x), B, C, H, W, float(eps), 0,
                      STREAM)
        need_affine = (weight is not None and weight.requires_grad) or (bias is not None and bias.requires_grad)
        ctx.save_for_backward(weight, running_mean, running_var, x if need_affine else None)
        ctx.eps = float(eps)
        return y

    @staticmethod
    def backward(ctx, gy):
        weight, running_mean, running_var, x = ctx.saved_tensors
        gy = _c(gy)
        B, C, H, W = gy.shape
        gx = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty_like(gy)
            engine().call("cc_bn_eval_fwd", gy, weight, None, running_mean, running_var, gx, _ws(8 * C, gy), B, C, H, W, ctx.eps, 1,
                          STREAM)
        gw, gb = bn_eval_affine_grads(gy, x, running_mean, running_var, ctx.eps, ctx.needs_input_grad[1], ctx.needs_input_grad[2])
        return gx, gw, gb, None, None, None


def bn_eval_affine_grads(gy, x, running_mean, running_var, eps, want_w=True, want_b=True):
    """d/d(weight, bias) of eval-mode BatchNorm  y = (x - mean) / sqrt(var + eps) * weight + bias  (fixed statistics): per-channel
    sums over (n, h, w).  Outside the training step (the step runs the layer in train() mode; train.py's --fix-* variants freeze
    the parameters): stock tensor reductions on the device."""
    gw = gb = None
    if want_w and x is not None:
        inv = torch.rsqrt(running_var + eps).view(1, -1, 1, 1)
        gw = (gy * ((x - running_mean.view(1, -1, 1, 1)) * inv)).sum(dim=(0, 2, 3))
    if want_b:
        gb = gy.sum(dim=(0, 2, 3))
    return gw, gb


def batch_norm(x, weight, bias, running_mean, running_var, num_batches_tracked, training, momentum, eps):
    """nn.BatchNorm2d.forward on csrc/bnorm.hip.  Training mode: three launches for >= 16 k values per channel, one
    workgroup-per-channel launch below.  Eval mode: the affine map of the running statistics (cc_bn_eval_fwd), differentiable
    w.r.t. the input and the affine parameters.  Outside the reference's use of the layer (4-d input, fixed momentum, tracked
    statistics) it raises."""
    if x.dim() != 4:
        raise NotImplementedError("ccengine BatchNorm: 4-d NCHW input (the reference's nn.BatchNorm2d use)")
    if not training:
        if running_mean is None or running_var is None:
            raise NotImplementedError("ccengine BatchNorm: eval mode needs tracked running statistics")
        return _BNEvalFn.apply(x, weight, bias, running_mean, running_var, eps)
    if momentum is None:
        raise NotImplementedError("ccengine BatchNorm: a fixed momentum (the reference's nn.BatchNorm2d use)")
    if x.shape[0] * x.shape[2] * x.shape[3] == 1:
        raise ValueError("Expected more than 1 value per channel when training, got input size %s" % (list(x.shape),))
    if num_batches_tracked is not None:
        num_batches_tracked.add_(1)
    return _BNTrainFn.apply(x, weight, bias, running_mean, running_var, momentum, eps)


class _Up2xFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, scale):
        x = _c(x)
        B, C, H, W = x.shape
        y = torch.empty(B, C, 2 * H, 2 * W, device=x.device, dtype=torch.float32)
        engine().call("cc_upsample2x_fwd", x, y, B, C, H, W, C * H * W, 4 * C * H * W, float(scale), STREAM)
        ctx.geom = (B, C, H, W, float(scale))
        return y

    @staticmethod
    def backward(ctx, gy):
        B, C, H, W, scale = ctx.geom
        gy, gy_bs = _slice_or_c(gy)
        gx = torch.empty(B, C, H, W, device=gy.device, dtype=torch.float32)
        engine().call("cc_upsample2x_bwd", gy, gx, B, C, H, W, gy_bs, C * H * W, scale, 0, STREAM)
        return gx, None


def upsample_bilinear2x(x, scale=1.0):
    """scale * F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=False) in one launch (csrc/resize.hip)."""
    return _Up2xFn.apply(x, float(scale))


# ----------------------------------------------------------------------------- cost volume
_perm_cache = {}


def _inv_perm(perm, device):
    key = (tuple(perm), str(device))
    t = _perm_cache.get(key)
    if t is None:
        inv = [0] * len(perm)
        for k, d in enumerate(perm):
            inv[d] = k                       # displacement d is stored in output channel k
        t = torch.tensor(inv, dtype=torch.int32, device=device)
        _perm_cache[key] = t
    return t


class _CorrPairFn(torch.autograd.Function):
    """cat(correlate(a, b)[:, perm_b], correlate(a, c)[:, perm_c]) -> [B,162,H,W] (back2future.py:173-177)."""

    @staticmethod
    def forward(ctx, a, b, c, inv_b, inv_c):
        a, b, c = _c(a), _c(b), _c(c)
        B, C, H, W = a.shape
        out = torch.empty(B, 162, H, W, device=a.device, dtype=torch.float32)
        E = engine()
        E.call("cc_corr9x9_fwd", a, b, out, inv_b, B, C, H, W, 162, 0, STREAM)
        E.call("cc_corr9x9_fwd", a, c, out, inv_c, B, C, H, W, 162, 81, STREAM)
        ctx.save_for_backward(a, b, c, inv_b, inv_c)
        return out

    @staticmethod
    def backward(ctx, g):
        a, b, c, inv_b, inv_c = ctx.saved_tensors
        g = _c(g)
        B, C, H, W = a.shape
        E = engine()
        ga = torch.empty_like(a)
        gb = torch.empty_like(b) if ctx.needs_input_grad[1] else None
        gc = torch.empty_like(c) if ctx.needs_input_grad[2] else None
        E.call("cc_corr9x9_bwd", g, a, b, ga, gb, inv_b, B, C, H, W, 162, 0, 0, STREAM)
        E.call("cc_corr9x9_bwd", g, a, c, ga, gc, inv_c, B, C, H, W, 162, 81, 1, STREAM)
        return ga, gb, gc, None, None


def correlation_pair(a, b, c, perm_b, perm_c):
    return _CorrPairFn.apply(a, b, c, _inv_perm(perm_b, a.device), _inv_perm(perm_c, a.device))


def correlate(input1, input2):
    """models/back2future.py:15-25 `correlate` (no permutation): [B,81,H,W]."""
    a, b = _c(input1), _c(input2)
    return _CorrFn.apply(a, b)


class _CorrFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        B, C, H, W = a.shape
        out = torch.empty(B, 81, H, W, device=a.device, dtype=torch.float32)
        engine().call("cc_corr9x9_fwd", a, b, out, None, B, C, H, W, 81, 0, STREAM)
        ctx.save_for_backward(a, b)
        return out

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        B, C, H, W = a.shape
        ga, gb = torch.empty_like(a), torch.empty_like(b)
        engine().call("cc_corr9x9_bwd", _c(g), a, b, ga, gb, None, B, C, H, W, 81, 0, 0, STREAM)
        return ga, gb


class _CorrPatchFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, patch, dilation):
        a, b = _c(a), _c(b)
        B, C, H, W = a.shape
        out = torch.empty(B, patch * patch, H, W, device=a.device, dtype=torch.float32)
        engine().call("cc_corr_patch_fwd", a, b, out, B, C, H, W, patch, dilation, STREAM)
        ctx.save_for_backward(a, b)
        ctx.geom = (patch, dilation)
        return out

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        B, C, H, W = a.shape
        ga = torch.empty_like(a) if ctx.needs_input_grad[0] else None
        gb = torch.empty_like(b) if ctx.needs_input_grad[1] else None
        engine().call("cc_corr_patch_bwd", _c(g), a, b, ga, gb, B, C, H, W, ctx.geom[0], ctx.geom[1], STREAM)
        return ga, gb, None, None


def correlate_patch(input1, input2, patch_size=21, dilation_patch=2):
    """models/FlowNetC6.py:18-30 `correlate`: [B, patch^2, H, W], already divided by C."""
    return _CorrPatchFn.apply(input1, input2, int(patch_size), int(dilation_patch))
