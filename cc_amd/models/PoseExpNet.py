"""models.PoseExpNet (reference models/PoseExpNet.py:18-94; BASELINE config 1) on the gfx950 conv kernels.
Returns (masks, pose) -- unlike the *B6 nets."""
import torch
import torch.nn as nn

from .. import nn as L
from ._blocks import xavier_zero_bias, seq_conv_act
from .PoseNetB6 import PLANES, KS


class PoseExpNet(nn.Module):
    def __init__(self, nb_ref_imgs=2, output_exp=False):
        super().__init__()
        self.nb_ref_imgs, self.output_exp = nb_ref_imgs, output_exp
        cin = 3 * (1 + nb_ref_imgs)
        for i in range(7):
            setattr(self, "conv%d" % (i + 1), seq_conv_act(cin, PLANES[i], KS[i], 2, "relu"))
            cin = PLANES[i]
        self.pose_pred = L.Conv2d(cin, 6 * nb_ref_imgs, 1, 1, 0)
        if output_exp:
            up = [256, 128, 64, 32, 16]
            ins = [PLANES[4]] + up[:4]
            for j, lvl in enumerate(range(5, 0, -1)):
                setattr(self, "upconv%d" % lvl, nn.Sequential(L.ConvTranspose2d(ins[j], up[j], 4, 2, 1, act="relu"), L.Act()))
            for j, lvl in enumerate(range(4, 0, -1)):
                setattr(self, "predict_mask%d" % lvl, L.Conv2d(up[j + 1], nb_ref_imgs, 3, 1, 1, act="sigmoid"))

    def init_weights(self):
        xavier_zero_bias(self)

    def forward(self, target_image, ref_imgs):
        assert len(ref_imgs) == self.nb_ref_imgs
        inp = torch.cat([target_image] + list(ref_imgs), 1)
        c = [inp]
        for i in range(7):
            c.append(getattr(self, "conv%d" % (i + 1))(c[-1]))
        p = self.pose_pred(c[7]).mean(3).mean(2)
        pose = 0.01 * p.view(p.size(0), self.nb_ref_imgs, 6)
        masks = {l: None for l in range(1, 5)}
        if self.output_exp:
            u = c[5]
            for lvl in range(5, 0, -1):
                ref = c[lvl - 1]
                u = getattr(self, "upconv%d" % lvl)(u)[:, :, 0:ref.size(2), 0:ref.size(3)]
                if lvl <= 4:
                    masks[lvl] = getattr(self, "predict_mask%d" % lvl)(u)
        if self.training:
            return [masks[1], masks[2], masks[3], masks[4]], pose
        return masks[1], pose
