"""models.MaskResNet6 (reference models/MaskResNet6.py:72-160): MaskNet6's decoder on a ResNet encoder -- a 7x7 stride-2 stem
and five 2-block BasicBlock stages (1x1 + BatchNorm shortcuts), 4x4 stride-2 deconvs, per-scale sigmoid mask heads."""
import torch
import torch.nn as nn

from .. import nn as L
from ._blocks import xavier_zero_bias, seq_conv_act, make_layer

_ENC = [16, 32, 64, 128, 256, 256]
_UP = [256, 256, 128, 64, 32, 16]


class MaskResNet6(nn.Module):
    def __init__(self, nb_ref_imgs=4, output_exp=True):
        super().__init__()
        self.nb_ref_imgs, self.output_exp = nb_ref_imgs, output_exp
        self.conv1 = seq_conv_act(3 * (1 + nb_ref_imgs), _ENC[0], 7, 2, "relu")
        for i in range(1, 6):
            setattr(self, "conv%d" % (i + 1), make_layer(_ENC[i - 1], _ENC[i], 2, 2))
        if output_exp:
            ins = [_ENC[5]] + [_UP[j] + _ENC[4 - j] for j in range(5)]
            for j, lvl in enumerate(range(6, 0, -1)):
                setattr(self, "deconv%d" % lvl, nn.Sequential(L.ConvTranspose2d(ins[j], _UP[j], 4, 2, 1, act="relu"), L.Act()))
            for j, lvl in enumerate(range(6, 0, -1)):
                setattr(self, "pred_mask%d" % lvl, L.Conv2d(_UP[j], nb_ref_imgs, 3, 1, 1, act="sigmoid"))

    def init_weights(self):
        xavier_zero_bias(self)

    def init_mask_weights(self):
        """MaskResNet6.py:107-119."""
        for m in self.modules():
            if isinstance(m, nn.ConvTranspose2d):
                nn.init.xavier_uniform_(m.weight.data)
                if m.bias is not None:
                    m.bias.data.zero_()
        for lvl in range(1, 7):
            m = getattr(self, "pred_mask%d" % lvl)
            nn.init.xavier_uniform_(m.weight.data)
            m.bias.data.zero_()

    def forward(self, target_image, ref_imgs):
        assert len(ref_imgs) == self.nb_ref_imgs
        x = torch.cat([target_image] + list(ref_imgs), 1)
        c = []
        for i in range(6):
            x = getattr(self, "conv%d" % (i + 1))(x)
            c.append(x)
        masks = {l: None for l in range(1, 7)}
        if self.output_exp:
            u = self.deconv6(c[5])
            masks[6] = self.pred_mask6(u)
            for lvl in range(5, 0, -1):
                u = getattr(self, "deconv%d" % lvl)(torch.cat((u, c[lvl - 1]), 1))
                masks[lvl] = getattr(self, "pred_mask%d" % lvl)(u)
        if self.training:
            return tuple(masks[l] for l in range(1, 7))
        return masks[1]
