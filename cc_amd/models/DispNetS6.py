"""models.DispNetS6 (reference models/DispNetS6.py:42-137): DispNetS with disparity heads at all six decoder scales."""
import torch
import torch.nn as nn

from .. import nn as L
from .. import ops
from ._blocks import xavier_zero_bias, crop_like

_ENC = [32, 64, 128, 256, 512, 512, 512]
_DEC = [512, 512, 256, 128, 64, 32, 16]
_K = [7, 5, 3, 3, 3, 3, 3]


def _down(cin, cout, k):
    p = (k - 1) // 2
    return nn.Sequential(L.Conv2d(cin, cout, k, 2, p, act="relu"), L.Act(), L.Conv2d(cout, cout, k, 1, p, act="relu"), L.Act())


class DispNetS6(nn.Module):
    def __init__(self, alpha=10, beta=0.01):
        super().__init__()
        self.alpha, self.beta = alpha, beta
        cin = 3
        for i in range(7):
            setattr(self, "conv%d" % (i + 1), _down(cin, _ENC[i], _K[i]))
            cin = _ENC[i]
        up_in = [_ENC[6]] + _DEC[:6]
        for j, lvl in enumerate(range(7, 0, -1)):
            setattr(self, "upconv%d" % lvl,
                    nn.Sequential(L.ConvTranspose2d(up_in[j], _DEC[j], 3, 2, 1, 1, act="relu"), L.Act()))
        skip = {7: _ENC[5], 6: _ENC[4], 5: _ENC[3], 4: _ENC[2], 3: 1 + _ENC[1], 2: 1 + _ENC[0], 1: 1}
        for j, lvl in enumerate(range(7, 0, -1)):
            setattr(self, "iconv%d" % lvl, nn.Sequential(L.Conv2d(_DEC[j] + skip[lvl], _DEC[j], 3, 1, 1, act="relu"), L.Act()))
        for lvl in range(6, 0, -1):
            setattr(self, "predict_disp%d" % lvl, nn.Sequential(L.Conv2d(_DEC[7 - lvl], 1, 3, 1, 1), L.Act()))

    def init_weights(self):
        xavier_zero_bias(self)

    def forward(self, x):
        c = [x]
        for i in range(1, 8):
            c.append(getattr(self, "conv%d" % i)(c[-1]))
        out, disps, prev = c[7], {}, None
        for lvl in range(7, 0, -1):
            skip = c[lvl - 1]
            up = crop_like(getattr(self, "upconv%d" % lvl)(out), skip)
            parts = [up] if lvl == 1 else [up, skip]
            if lvl <= 3:
                parts.append(crop_like(ops.upsample_bilinear2x(prev), skip))
            out = getattr(self, "iconv%d" % lvl)(torch.cat(parts, 1))
            if lvl <= 6:
                head = getattr(self, "predict_disp%d" % lvl)[0]
                prev = ops.conv2d(out, head.weight, head.bias, 1, 1, "sigmoid", None, float(self.alpha), float(self.beta))
                disps[lvl] = prev
        if self.training:
            return tuple(disps[l] for l in range(1, 7))
        return disps[1]
