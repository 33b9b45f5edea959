"""models.PoseNetB6 (reference models/PoseNetB6.py:24-83) on the gfx950 conv kernels."""
import torch
import torch.nn as nn

from .. import nn as L
from ._blocks import xavier_zero_bias, seq_conv_act, conv_call

PLANES = [16, 32, 64, 128, 256, 256, 256, 256]
KS = [7, 5, 3, 3, 3, 3, 3, 3]


class PoseNetB6(nn.Module):
    def __init__(self, nb_ref_imgs=2):
        super().__init__()
        self.nb_ref_imgs = nb_ref_imgs
        cin = 3 * (1 + nb_ref_imgs)
        for i in range(8):
            setattr(self, "conv%d" % (i + 1), seq_conv_act(cin, PLANES[i], KS[i], 2, "relu"))
            cin = PLANES[i]
        self.pose_pred = L.Conv2d(cin, 6 * nb_ref_imgs, 1, 1, 0)

    def init_weights(self):
        xavier_zero_bias(self)

    def forward(self, target_image, ref_imgs):
        assert len(ref_imgs) == self.nb_ref_imgs
        x = torch.cat([target_image] + list(ref_imgs), 1)       # target first (PoseNetB6.py:67-69)
        for i in range(8):      # a pure conv -> conv chain: every ReLU backward runs in the next layer's data-gradient epilogue
            x = conv_call(getattr(self, "conv%d" % (i + 1))[0], x, pre_act="relu" if i else None, defer=True)
        pose = conv_call(self.pose_pred, x, pre_act="relu").mean(3).mean(2)       # W first, then H (PoseNetB6.py:80)
        return 0.01 * pose.view(pose.size(0), self.nb_ref_imgs, 6)
