"""models.PoseNetB6 (reference models/PoseNetB6.py:24-83) on the gfx950 conv kernels."""
import torch
import torch.nn as nn

from .. import nn as L
from ..tape import run_network
from ._blocks import xavier_zero_bias, seq_conv_act

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

    def _body(self, tape, x):
        # a pure conv -> conv chain: every ReLU backward runs in the next layer's data-gradient epilogue
        for i in range(8):
            m = getattr(self, "conv%d" % (i + 1))[0]
            x = tape.conv(x, m.weight, m.bias, 2, m.padding[0], "relu")
        p = tape.conv(x, self.pose_pred.weight, self.pose_pred.bias, 1, 0, None)
        n = self.nb_ref_imgs
        # W first, then H (PoseNetB6.py:80); [B, 6n] values: stock torch
        return tape.torch_fn([p], lambda t: (0.01 * t.mean(3).mean(2).view(t.size(0), n, 6),))

    def forward(self, target_image, ref_imgs):
        assert len(ref_imgs) == self.nb_ref_imgs
        x = torch.cat([target_image] + list(ref_imgs), 1)       # target first (PoseNetB6.py:67-69)
        return run_network(self._body, [x], list(self.parameters()))[0]
