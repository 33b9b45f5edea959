"""models.DispResNet6 (reference models/DispResNet6.py:97-194) on the gfx950 conv kernels."""
import torch
import torch.nn as nn

from .. import nn as L
from .. import ops
from ..tape import run_network
from ._blocks import xavier_zero_bias, make_layer, basic_block

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

    # Gradient chunks for a trainer that exchanges / updates a network's parameters while its backward pass still runs
    # (cc_amd/trainer.py, per-network pipeline): tag of a tape.mark() in _body -> the first parameter (registration order = the order
    # of the optimizer's bucket) whose gradient is complete when the backward pass reaches that mark; everything from there to the
    # previously completed part is final.  The backward pass walks the decoder first, then conv7 ... conv1.
    GRAD_CHUNKS = (("decoder", "upconv7.0.weight"), ("conv5", "conv5.0.conv1.weight"))

    def _body(self, tape, x):
        """The forward pass on the tape: every decoder concatenation (upconv, skip[, disp_up]) is one buffer whose slices the
        up-convolution, the encoder stage and the disparity up-sampling write directly."""
        B, _, H, W = x.t.shape
        hw = [(H, W)]
        for _ in range(7):
            hw.append(((hw[-1][0] - 1) // 2 + 1, (hw[-1][1] - 1) // 2 + 1))          # k7/s2/p3 and k3/s2/p1 alike
        cats = {}
        for j, lvl in enumerate(range(7, 0, -1)):
            chans = [_DEC[j]] + ([_ENC[lvl - 2]] if lvl > 1 else []) + ([1] if lvl <= 3 else [])
            cats[lvl] = tape.concat(B, chans, hw[lvl - 1][0], hw[lvl - 1][1], x.t)
        c1a = tape.conv(x, self.conv1[0].weight, self.conv1[0].bias, 2, 3, "relu")
        c = [x, tape.conv(c1a, self.conv1[2].weight, self.conv1[2].bias, 1, 3, "relu", out=cats[2].slot(1))]
        for i in range(2, 8):
            if i == 5:
                tape.mark("conv5")         # backward: conv5 .. conv7 (27.5 M of the 54.6 M parameters) are done here
            stage = getattr(self, "conv%d" % i)
            y = tape.tap(basic_block(tape, stage[0], c[-1]), "conv%d.0" % i)
            c.append(tape.tap(basic_block(tape, stage[1], y, out=cats[i + 1].slot(1) if i < 7 else None), "conv%d.1" % i))
        tape.mark("decoder")               # backward: the whole decoder (24.3 M parameters) is done here
        out, disps, prev = c[7], {}, None
        for lvl in range(7, 0, -1):
            cb = cats[lvl]
            Hs, Ws = hw[lvl - 1]
            up_m = getattr(self, "upconv%d" % lvl)[0]
            fits = (2 * out.t.shape[2], 2 * out.t.shape[3]) == (Hs, Ws)
            up = tape.conv_transpose(out, up_m.weight, up_m.bias, 2, 1, 1, "relu", out=cb.slot(0) if fits else None)
            cb.put(0, tape.crop(up, Hs, Ws))
            if lvl > 1:
                cb.put(1, c[lvl - 1])
            if lvl <= 3:
                k = 2 if lvl > 1 else 1
                fits = (2 * prev.t.shape[2], 2 * prev.t.shape[3]) == (Hs, Ws)
                du = tape.upsample2x(prev, 1.0, out=cb.slot(k) if fits else None)
                cb.put(k, tape.crop(du, Hs, Ws))
            out = basic_block(tape, getattr(self, "iconv%d" % lvl)[0], cb.done())
            if lvl <= 6:
                head = getattr(self, "predict_disp%d" % lvl)[0]
                prev = tape.conv(out, head.weight, head.bias, 1, 1, "sigmoid", float(self.alpha), float(self.beta))
                disps[lvl] = prev
        return [disps[l] for l in range(1, 7)] if self.training else [disps[1]]

    def forward(self, x):
        outs = run_network(self._body, [x], list(self.parameters()))
        return tuple(outs) if self.training else outs[0]
