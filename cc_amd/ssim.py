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


def _ssim_any_channels(img1, img2):
    """The 13-tap HIP kernels take planes in groups of three (the step's RGB frames); the map is per channel (depth-wise window), so any
    channel count is the same computation on [B * C] planes, padded with zero planes to a multiple of three."""
    B, C, H, W = img1.shape
    a, b = img1.contiguous().float().reshape(B * C, H, W), img2.contiguous().float().reshape(B * C, H, W)
    pad = (-B * C) % 3
    if pad:
        z = a.new_zeros(pad, H, W)
        a, b = torch.cat([a, z]), torch.cat([b, z])
    out = _SSIMFn.apply(a.reshape(-1, 3, H, W), b.reshape(-1, 3, H, W)).reshape(-1, H, W)
    return out[:B * C].reshape(B, C, H, W)


def _ssim_generic(img1, img2, window_size):
    """ssim.py:19-36 for window sizes other than the 13 taps the training losses use (the module form defaults to 11): depth-wise
    Gaussian windows as stock grouped convolutions + the SSIM algebra.  NOT on the training step's path (loss_functions.py:80 calls
    ssim(..., window_size 13)): a generality fallback that runs on the vendor convolution library."""
    import torch.nn.functional as F
    C = img1.size(1)
    window = create_window(window_size, C).to(device=img1.device, dtype=img1.dtype)
    pad = window_size // 2
    mu1, mu2 = F.conv2d(img1, window, padding=pad, groups=C), F.conv2d(img2, window, padding=pad, groups=C)
    mu1_sq, mu2_sq, mu1_mu2 = mu1.pow(2), mu2.pow(2), mu1 * mu2
    sigma1_sq = F.conv2d(img1 * img1, window, padding=pad, groups=C) - mu1_sq
    sigma2_sq = F.conv2d(img2 * img2, window, padding=pad, groups=C) - mu2_sq
    sigma12 = F.conv2d(img1 * img2, window, padding=pad, groups=C) - mu1_mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    return ((2 * mu1_mu2 + C1) * (2 * sigma12 + C2)) / ((mu1_sq + mu2_sq + C1) * (sigma1_sq + sigma2_sq + C2))


def ssim(img1, img2, window_size=13, size_average=True):
    """ssim.py:68-76: the un-reduced per-channel SSIM map (size_average is ignored there too).  Window 13 (the reference's default
    and what its losses use) runs on the HIP kernels for any channel count; other window sizes take the generic path."""
    if window_size != WINDOW_SIZE:
        return _ssim_generic(img1, img2, window_size)
    if img1.size(1) != 3:
        return _ssim_any_channels(img1, img2)
    return _SSIMFn.apply(img1, img2)


class SSIM(torch.nn.Module):
    """ssim.py:42-66 (module form; its default window is 11, as in the reference)."""

    def __init__(self, window_size=11, size_average=True):
        super().__init__()
        self.window_size = window_size
        self.size_average = size_average

    def forward(self, img1, img2):
        return ssim(img1, img2, self.window_size, self.size_average)
