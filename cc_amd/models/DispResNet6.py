"""models.DispResNet6 (reference models/DispResNet6.py:97-194) on the gfx950 conv kernels."""
import torch
import torch.nn as nn

from .. import nn as L
from .. import ops
from ._blocks import xavier_zero_bias, crop_like, make_layer, conv_call

_ENC = [32, 64, 128, 256, 512, 512, 512]
_DEC = [512, 512, 256, 128, 64, 32, 16]


class DispResNet6(nn.Module):
    """Same constructor, state_dict keys (conv1.0.weight ... predict_disp1.0.bias), train()/eval() outputs."""

    def __init__(self, alpha=10, beta=0.01):
        super().__init__()
        self.alpha, self.beta = alpha, beta
        self.conv1 = nn.Sequential(L.Conv2d(3, _ENC[0], 7, 2, 3, act="relu"), L.Act(),
                                   L.Conv2d(_ENC[0], _ENC[0], 7, 1, 3, act="relu"), L.Act())
        for i in range(1, 7):
            setattr(self, "conv%d" % (i + 1), make_layer(_ENC[i - 1], _ENC[i], 2, 2))
        up_in = [_ENC[6]] + _DEC[:6]
        for j, lvl in enumerate(range(7, 0, -1)):
            setattr(self, "upconv%d" % lvl,
                    nn.Sequential(L.ConvTranspose2d(up_in[j], _DEC[j], 3, 2, 1, 1, act="relu"), L.Act()))
        skip = {7: _ENC[5], 6: _ENC[4], 5: _ENC[3], 4: _ENC[2], 3: 1 + _ENC[1], 2: 1 + _ENC[0], 1: 1}
        for j, lvl in enumerate(range(7, 0, -1)):
            setattr(self, "iconv%d" % lvl, make_layer(_DEC[j] + skip[lvl], _DEC[j], 1, 1))
        for lvl in range(6, 0, -1):
            # Conv3x3(->1) + Sigmoid; alpha * sigmoid + beta is fused into the conv epilogue
            setattr(self, "predict_disp%d" % lvl, nn.Sequential(L.Conv2d(_DEC[7 - lvl], 1, 3, 1, 1), L.Act()))

    def init_weights(self):
        xavier_zero_bias(self)

    def _disp(self, lvl, feat):
        head = getattr(self, "predict_disp%d" % lvl)[0]
        return ops.conv2d(feat, head.weight, head.bias, 1, 1, "sigmoid", None, float(self.alpha), float(self.beta))

    def forward(self, x):
        c = [x, conv_call(self.conv1[2], conv_call(self.conv1[0], x, defer=True), pre_act="relu")]
        for i in range(2, 8):
            c.append(getattr(self, "conv%d" % i)(c[-1]))
        out, disps, prev = c[7], {}, None
        for lvl in range(7, 0, -1):
            skip = c[lvl - 1]
            up = crop_like(getattr(self, "upconv%d" % lvl)(out), skip)
            parts = [up] if lvl == 1 else [up, skip]                      # concat order (upconv, skip[, disp_up])
            if lvl <= 3:
                parts.append(crop_like(ops.upsample_bilinear2x(prev), skip))
            out = getattr(self, "iconv%d" % lvl)(torch.cat(parts, 1))
            if lvl <= 6:
                prev = self._disp(lvl, out)
                disps[lvl] = prev
        if self.training:
            return tuple(disps[l] for l in range(1, 7))
        return disps[1]
