"""Heterogeneous convolution launch lists (cc_conv2d_list, include/ccengine.h): independent convolution problems of
different layers / networks packed into as few kernel launches as their tile configurations allow.

A problem is described once (static shapes, static buffers) as a 32-long record; a list of records is one C-ABI call.
The reference runs its four networks one after the other (train.py:454-463) although they only share their input; on a
256-CU device the deep, small-map layers of one network cannot fill the chip alone, the same layers next to another
network's large-map layers can.

Probe code (tools/merge_probe.py, one parity test): the measurement said packing does NOT pay on this chip
(profiles/r03_merge_probe.txt), so the training step does not use it and the module lives here, not in the package."""
import ctypes
import struct

import torch

from cc_amd._lib import engine, STREAM
from cc_amd import ops

CL_LONGS = 32


def _fbits(v):
    return struct.unpack("<I", struct.pack("<f", float(v)))[0]


def _p(t):
    return 0 if t is None else t.data_ptr()


def _bs(t):
    """batch stride (elements) of an NCHW tensor or channel-slice view"""
    return 0 if t is None else (t.stride(0) if t.dim() == 4 else 0)


def conv_record(x, w, bias, res, y, stride, pad, act=0, act_a=1.0, act_b=0.0):
    """conv2d forward arithmetic: y = act(conv2d(x, w) + bias + res); x / y / res may be channel slices of wider tensors."""
    B, Cin, IH, IW = x.shape
    Cout, _, R, S = w.shape
    OH, OW = y.shape[2], y.shape[3]
    geom = (B, Cin, IH, IW, Cout, R, S, stride, pad, OH, OW)
    pk = ops.packs.ensure("fwd", w, geom)
    if pk is None:
        return None
    return [0, _p(x), _p(w), _p(bias), _p(res), _p(y), _p(pk), 0,
            B, Cin, IH, IW, _bs(x), Cout, R, S, stride, pad, OH, OW, _bs(y), _bs(res), 0,
            act, _fbits(act_a), _fbits(act_b), 0, 0, 0, 0, 0, 0]


def tconv_record(gy, w, bias, gx, stride, pad, w_k_stride, w_c_stride, R, S, act=0, act_a=1.0, act_b=0.0, mul=None, add=None):
    """transposed arithmetic gx[n,c,iy,ix] = sum_{k,r,s} w(k,c,r,s) gy[n,k,oy,ox] (iy = oy*stride - pad + r): the data-gradient
    of conv2d and the forward of ConvTranspose2d.  mul: gx = (sum + add) * act'(mul) (act describes the layer whose OUTPUT mul
    is); without mul: gx = act(sum + bias)."""
    B, K, OH, OW = gy.shape
    C, IH, IW = gx.shape[1], gx.shape[2], gx.shape[3]
    geom = (B, K, OH, OW, C, R, S, stride, pad, IH, IW, w_k_stride, w_c_stride)
    pk = ops.packs.ensure("dgrad", w, geom)
    if pk is None:
        return None
    assert add is None or mul is not None
    return [1, _p(gy), _p(w), _p(bias), _p(mul), _p(gx), _p(pk), _p(add),
            B, K, OH, OW, _bs(gy), C, R, S, stride, pad, IH, IW, _bs(gx), _bs(mul), _bs(add),
            act, _fbits(act_a), _fbits(act_b), w_k_stride, w_c_stride, 0, 0, 0, 0]


class LaunchList:
    """n records -> one cc_conv2d_list call; the host array and the workspace are built once."""

    def __init__(self, records, device, split_target=0):
        assert records and all(r is not None and len(r) == CL_LONGS for r in records)
        self.n = len(records)
        flat = [int(v) for r in records for v in r]
        self.host = (ctypes.c_long * len(flat))(*flat)
        self.split_target = int(split_target)
        E = engine()
        nbytes = E.call("cc_conv2d_list_ws_bytes", self.n, ctypes.addressof(self.host), self.split_target)
        assert nbytes > 0, "cc_conv2d_list: malformed record"
        self.ws = torch.empty(max(nbytes // 4, 64), device=device, dtype=torch.float32)

    def run(self):
        engine().call("cc_conv2d_list", self.n, ctypes.addressof(self.host), self.ws, self.split_target, STREAM)
