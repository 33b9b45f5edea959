"""models.PoseNet6 (reference models/PoseNet6.py:20-62): PoseNetB6's encoder behind an extra stride-2 3x3 stem (conv0) that
keeps the 3*(1+nb_ref) input channels; 7 further stride-2 convs, 1x1 pose head, spatial mean, x0.01."""
import torch
import torch.nn as nn

from .. import nn as L
from ._blocks import xavier_zero_bias, seq_conv_act

PLANES = [16, 32, 64, 128, 256, 256, 256]
KS = [7, 5, 3, 3, 3, 3, 3]


class PoseNet6(nn.Module):
    def __init__(self, nb_ref_imgs=2):
        super().__init__()
        self.nb_ref_imgs = nb_ref_imgs
        cin = 3 * (1 + nb_ref_imgs)
        self.conv0 = seq_conv_act(cin, cin, 3, 2, "relu")                   # PoseNet6.py:27
        for i in range(7):
            setattr(self, "conv%d" % (i + 1), seq_conv_act(cin, PLANES[i], KS[i], 2, "relu"))
            cin = PLANES[i]
        self.pose_pred = L.Conv2d(cin, 6 * nb_ref_imgs, 1, 1, 0)

    def init_weights(self):
        xavier_zero_bias(self)

    def forward(self, target_image, ref_imgs):
        assert len(ref_imgs) == self.nb_ref_imgs
        x = torch.cat([target_image] + list(ref_imgs), 1)
        x = self.conv0(x)
        for i in range(7):
            x = getattr(self, "conv%d" % (i + 1))(x)
        pose = self.pose_pred(x).mean(3).mean(2)                             # PoseNet6.py:59
        return 0.01 * pose.view(pose.size(0), self.nb_ref_imgs, 6)
