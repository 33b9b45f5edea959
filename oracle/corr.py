"""TEST INFRASTRUCTURE ONLY (oracle) -- CPU restatement of the third-party
``spatial_correlation_sampler.spatial_correlation_sample`` op.

The op is NOT in /root/reference (requirements.txt:13, unpinned PyPI package
``spatial-correlation-sampler``, upstream ClementPinard/Pytorch-Correlation-
extension).  Restated from its published semantic and the reference's call
sites (models/back2future.py:15-25: kernel_size=1, patch_size=9, stride=1;
models/FlowNetC6.py:18-30: patch 21, dilation_patch 2):

    out[b, i, j, y, x] = sum_c in1[b, c, y, x] * in2[b, c, y + (i-r)*dp, x + (j-r)*dp]

with r = (patch-1)//2 and zero contribution outside in2.  **Parity unpinned**:
the reference carries no test/golden vector for this op, so this restatement is
the oracle by definition.
"""
import torch
import torch.nn.functional as F


def correlation_volume(in1, in2, patch_size=9, dilation_patch=1):
    """-> [B, patch, patch, H, W]; differentiable torch ops only."""
    B, C, H, W = in1.shape
    r = (patch_size - 1) // 2
    pad = r * dilation_patch
    in2p = F.pad(in2, (pad, pad, pad, pad))
    rows = []
    for i in range(patch_size):
        dy = i * dilation_patch
        cols = []
        for j in range(patch_size):
            dx = j * dilation_patch
            cols.append((in1 * in2p[:, :, dy:dy + H, dx:dx + W]).sum(1))
        rows.append(torch.stack(cols, 1))
    return torch.stack(rows, 1)


def correlate9(in1, in2):
    """models/back2future.py:15-25 ``correlate``: 81 channels, divided by C."""
    v = correlation_volume(in1, in2, 9, 1)
    b, ph, pw, h, w = v.shape
    return v.reshape(b, ph * pw, h, w) / in1.size(1)
