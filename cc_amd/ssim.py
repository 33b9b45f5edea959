"""Drop-in for the reference's ``ssim`` module (ssim.py) on the fused gfx950 SSIM kernels
(cc_amd/csrc/ssim.hip): one LDS-tiled separable 13-tap pass instead of five depth-wise 13x13
``conv2d`` calls; the window is built once (the reference rebuilds it on every call, ssim.py:70).
"""
import ctypes
from math import exp

import torch

from ._lib import engine, STREAM

WINDOW_SIZE = 13


def gaussian(window_size, sigma):
    """ssim.py:9-11."""
    gauss = torch.Tensor([exp(-(x - window_size // 2) ** 2 / float(2 * sigma ** 2)) for x in range(window_size)])
    return gauss / gauss.sum()


def create_window(window_size, channel):
    """ssim.py:13-17 (kept for API parity; the kernels use the separable 1-D taps)."""
    w1 = gaussian(window_size, 1.5).unsqueeze(1)
    w2 = w1.mm(w1.t()).float().unsqueeze(0).unsqueeze(0)
    return w2.expand(channel, 1, window_size, window_size).contiguous()


_taps = None


def gauss13_ptr():
    """Host address of the 13 fp32 taps (the one host pointer of the C ABI)."""
    global _taps
    if _taps is None:
        g = gaussian(WINDOW_SIZE, 1.5)
        _taps = (ctypes.c_float * WINDOW_SIZE)(*[float(v) for v in g])
    return ctypes.addressof(_taps)


class _SSIMFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img1, img2):
        img1, img2 = img1.contiguous().float(), img2.contiguous().float()
        B, C, H, W = img1.shape
        out = torch.empty_like(img1)
        engine().call("cc_ssim_fwd", img1, img2, out, gauss13_ptr(), B, H, W, STREAM)
        ctx.save_for_backward(img1, img2)
        return out

    @staticmethod
    def backward(ctx, gout):
        img1, img2 = ctx.saved_tensors
        B, C, H, W = img1.shape
        gout = gout.contiguous().float()
        E = engine()
        sa, sb, sc = torch.empty_like(img1), torch.empty_like(img1), torch.empty_like(img1)
        g1 = g2 = None
        if ctx.needs_input_grad[1]:
            g2 = torch.empty_like(img2)
            E.call("cc_ssim_bwd", img1, img2, gout, sa, sb, sc, g2, gauss13_ptr(), B, H, W, STREAM)
        if ctx.needs_input_grad[0]:
            g1 = torch.empty_like(img1)
            E.call("cc_ssim_bwd", img2, img1, gout, sa, sb, sc, g1, gauss13_ptr(), B, H, W, STREAM)
        return g1, g2


def ssim(img1, img2, window_size=13, size_average=True):
    """ssim.py:68-76: the un-reduced per-channel SSIM map (size_average is ignored there too)."""
    if window_size != WINDOW_SIZE:
        raise NotImplementedError("the HIP SSIM kernel is specialised for the 13-tap window the reference uses")
    if img1.size(1) != 3:
        raise NotImplementedError("the HIP SSIM kernel is specialised for 3-channel images")
    return _SSIMFn.apply(img1, img2)


class SSIM(torch.nn.Module):
    """ssim.py:42-66 (module form; window 13 only)."""

    def __init__(self, window_size=13, size_average=True):
        super().__init__()
        self.window_size = window_size
        self.size_average = size_average

    def forward(self, img1, img2):
        return ssim(img1, img2, self.window_size, self.size_average)
