"""ctypes binding of libccengine.so -- the C-ABI boundary (include/ccengine.h).

The signatures are parsed from the header itself, so the header is the single
source of truth.  There is NO fallback: if the shared library is missing or a
tensor is not a contiguous fp32 tensor on a HIP device, the call raises.
"""
import ctypes
import os
import re

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(_HERE), "include", "ccengine.h")
LIB_PATH = os.path.join(_HERE, "libccengine.so")

_CT = {"long": ctypes.c_long, "int": ctypes.c_int, "float": ctypes.c_float, "size_t": ctypes.c_size_t, "double": ctypes.c_double}

STREAM = object()      # placeholder argument: "the current torch stream"


def parse_header(path=HEADER):
    """-> {name: (restype, [argtypes], [arg names])} for every `cc_*` declaration."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(int|size_t|void)\s+(cc_\w+)\s*\(([^)]*)\)\s*;", src):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        types, names = [], []
        if args and args != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                if "*" in a:
                    types.append(ctypes.c_void_p)
                else:
                    base = a.replace("const ", "").split(" ")[0]
                    types.append(_CT[base])
                names.append(a.split(" ")[-1].lstrip("*"))
        out[name] = ({"int": ctypes.c_int, "size_t": ctypes.c_size_t, "void": None}[ret], types, names)
    return out


def image_dense(t):
    """True for a 4-D NCHW tensor whose images are dense ([C,H,W] contiguous) but whose batch stride may be wider: a channel
    slice of a concat buffer / of a concat gradient.  Entry points that take a batch stride (x_bs, gy_bs, a_bs ...) read such
    views in place."""
    if t.dim() != 4:
        return False
    B, C, H, W = t.shape
    st = t.stride()
    return st[3] == 1 and st[2] == W and st[1] == H * W and (B == 1 or st[0] >= C * H * W)


class Engine:
    def __init__(self, path=None, require_device=True):
        path = path or LIB_PATH
        if not os.path.isfile(path):
            raise RuntimeError(
                "ccengine: %s not found -- build it with `python -m cc_amd.build` (hipcc, gfx950). "
                "There is no CPU fallback." % path)
        self.path = path
        self.require_device = require_device
        self.lib = ctypes.CDLL(path)
        self.sigs = parse_header()
        self.fn = {}
        self.strided_ok = {}
        for name, (ret, types, names) in self.sigs.items():
            f = getattr(self.lib, name)      # AttributeError if the library lacks a declared symbol
            f.restype = ret
            f.argtypes = types
            self.fn[name] = f
            # per-image dense channel slices (wider batch stride) are accepted only by entry points that TAKE batch strides /
            # channel totals; everywhere else a non-contiguous tensor is an error, not a silently mis-strided read
            self.strided_ok[name] = any(n.endswith("_bs") or "bstride" in n or n.endswith("channels_total") for n in names)

    def _ptr(self, t, name, i):
        if t is None:
            return None
        if not torch.is_tensor(t):
            raise TypeError("%s arg %d: expected a tensor or None, got %r" % (name, i, type(t)))
        if self.require_device and not t.is_cuda:
            raise RuntimeError("%s arg %d: tensor is on %s; the ccengine kernels need a HIP device (no CPU path)"
                               % (name, i, t.device))
        if t.dtype not in (torch.float32, torch.int32, torch.uint8, torch.int64, torch.float64):
            raise TypeError("%s arg %d: unsupported dtype %s" % (name, i, t.dtype))
        if not t.is_contiguous() and not (self.strided_ok.get(name, False) and image_dense(t)):
            raise ValueError("%s arg %d: tensor must be contiguous (per-image dense NCHW channel slices only where the entry "
                             "point takes their batch stride)" % (name, i))
        return t.data_ptr()

    def stream_ptr(self):
        if self.require_device:
            return torch.cuda.current_stream().cuda_stream
        return None

    def call(self, name, *args):
        f = self.fn[name]
        types = self.sigs[name][1]
        if len(args) != len(types):
            raise TypeError("%s expects %d arguments, got %d" % (name, len(types), len(args)))
        conv = []
        for i, (a, ty) in enumerate(zip(args, types)):
            if a is STREAM:
                conv.append(self.stream_ptr())
            elif ty is ctypes.c_void_p:
                # ints are raw addresses (only used for the documented host pointer gauss13_host)
                conv.append(a if isinstance(a, int) else self._ptr(a, name, i))
            else:
                conv.append(a)
        r = f(*conv)
        if self.sigs[name][0] is ctypes.c_int and r != 0:
            raise RuntimeError("%s failed with code %d" % (name, r))
        return r


_engine = None


def engine():
    global _engine
    if _engine is None:
        from . import config
        _engine = Engine(config.debug.library_path)       # None -> the product library next to this file
    return _engine


class use_library(object):
    """Context manager: run the enclosed calls on another build of the library (the tools build with its switches and timing
    registry -- cc_amd/build.py build_tools() -- for bench.py's instrumented step and the A/B scripts).  Not used by the product."""

    def __init__(self, path):
        self.e = Engine(path)

    def __enter__(self):
        global _engine
        self.prev = _engine
        _engine = self.e
        return self.e

    def __exit__(self, *a):
        global _engine
        _engine = self.prev
        return False


def _set_engine_for_tests(e):
    """Test hook (tests/hipemu): inject an Engine bound to the x86 emulation build of the SAME kernel
    sources.  Never called by the package itself."""
    global _engine
    _engine = e
