"""models.MaskNet6 (reference models/MaskNet6.py:19-123) on the gfx950 conv kernels."""
import torch
import torch.nn as nn

from .. import nn as L
from .. import ops
from ..tape import run_network
from ._blocks import xavier_zero_bias, seq_conv_act
from .PoseNetB6 import PLANES, KS

_UP = [256, 256, 128, 64, 32, 16]


class MaskNet6(nn.Module):
    def __init__(self, nb_ref_imgs=4, output_exp=True):
        super().__init__()
        self.nb_ref_imgs, self.output_exp = nb_ref_imgs, output_exp
        cin = 3 * (1 + nb_ref_imgs)
        for i in range(6):
            setattr(self, "conv%d" % (i + 1), seq_conv_act(cin, PLANES[i], KS[i], 2, "relu"))
            cin = PLANES[i]
        if output_exp:
            ins = [PLANES[5]] + [_UP[j] + PLANES[4 - j] for j in range(5)]
            for j, lvl in enumerate(range(6, 0, -1)):
                setattr(self, "deconv%d" % lvl, nn.Sequential(L.ConvTranspose2d(ins[j], _UP[j], 4, 2, 1, act="relu"), L.Act()))
            for j, lvl in enumerate(range(6, 0, -1)):
                setattr(self, "pred_mask%d" % lvl, L.Conv2d(_UP[j], nb_ref_imgs, 3, 1, 1, act="sigmoid"))

    def init_weights(self):
        xavier_zero_bias(self)

    def init_mask_weights(self):
        """MaskNet6.py:61-77."""
        for m in self.modules():
            if isinstance(m, nn.ConvTranspose2d):
                nn.init.xavier_uniform_(m.weight.data)
                if m.bias is not None:
                    m.bias.data.zero_()
        for lvl in range(1, 7):
            m = getattr(self, "pred_mask%d" % lvl)
            nn.init.xavier_uniform_(m.weight.data)
            m.bias.data.zero_()

    def _body(self, tape, x):
        """Encoder outputs and deconvolution outputs are written straight into the (out_upconv, out_conv) concatenation
        buffers of the decoder (MaskNet6.py:98-118); each prediction head is recorded before the concatenation that also
        consumes its input, so that the head's data-gradient is the last contribution and applies relu'."""
        B, _, H, W = x.t.shape
        hw = [(H, W)]
        for i in range(6):
            k, p = KS[i], (KS[i] - 1) // 2
            hw.append(((hw[-1][0] + 2 * p - k) // 2 + 1, (hw[-1][1] + 2 * p - k) // 2 + 1))
        cats = {}
        if self.output_exp:
            for j, lvl in enumerate(range(5, 0, -1)):                     # input of deconv{lvl}: (u_{lvl+1}, c[lvl-1])
                cats[lvl] = tape.concat(B, [_UP[j], PLANES[lvl - 1]], hw[lvl][0], hw[lvl][1], x.t)
        c = []
        for i in range(6):
            m = getattr(self, "conv%d" % (i + 1))[0]
            x = tape.conv(x, m.weight, m.bias, 2, m.padding[0], "relu", out=cats[i + 1].slot(1) if (i + 1) in cats else None)
            c.append(x)
        masks = {}
        if self.output_exp:
            u = None
            for lvl in range(6, 0, -1):
                d = getattr(self, "deconv%d" % lvl)[0]
                if lvl == 6:
                    src = c[5]
                else:
                    cb = cats[lvl]
                    cb.put(0, u)
                    cb.put(1, c[lvl - 1])
                    src = cb.done()
                nxt = cats.get(lvl - 1)
                OH, OW = 2 * src.t.shape[2], 2 * src.t.shape[3]
                fits = nxt is not None and tuple(nxt.slot(0).shape[2:]) == (OH, OW)
                u = tape.conv_transpose(src, d.weight, d.bias, 2, 1, 0, "relu", out=nxt.slot(0) if fits else None)
                pm = getattr(self, "pred_mask%d" % lvl)
                masks[lvl] = tape.conv(u, pm.weight, pm.bias, 1, 1, "sigmoid")
        if not self.output_exp:
            return []
        return [masks[l] for l in range(1, 7)] if self.training else [masks[1]]

    def forward(self, target_image, ref_imgs):
        assert len(ref_imgs) == self.nb_ref_imgs
        x = torch.cat([target_image] + list(ref_imgs), 1)
        if not self.output_exp:
            return tuple([None] * 6) if self.training else None
        outs = run_network(self._body, [x], list(self.parameters()))
        return tuple(outs) if self.training else outs[0]
