"""models.FlowNetC6 (reference models/FlowNetC6.py:32-164, submodules.py:5-41): FlowNetC with a 21x21 / dilation-2 cost volume on
the 1/8-resolution features, LeakyReLU(0.1) everywhere, six flow predictions (x div_flow, bilinearly up-sampled x2 when
full_res).  forward(x1, x2).  Same state_dict keys (conv1.0.weight ... upsampled_flow2_to_1.bias)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import nn as L
from .. import ops

_S = 0.1            # negative slope of every LeakyReLU in the net


def _conv(cin, cout, k=3, stride=1):
    """submodules.conv(batchNorm=False, ...): Conv2d + LeakyReLU(0.1) (parameters at index 0)."""
    return nn.Sequential(L.Conv2d(cin, cout, k, stride, (k - 1) // 2, act="lrelu", slope=_S), L.Act())


def _deconv(cin, cout):
    """submodules.deconv: ConvTranspose2d(4, 2, 1) + LeakyReLU(0.1)."""
    return nn.Sequential(L.ConvTranspose2d(cin, cout, 4, 2, 1, act="lrelu", slope=_S), L.Act())


def correlate(input1, input2):
    """FlowNetC6.py:18-30."""
    return ops.correlate_patch(input1, input2, 21, 2)


class FlowNetC6(nn.Module):
    def __init__(self, nlevels=5, batchNorm=False, div_flow=20, full_res=True, pretrained=True):
        super().__init__()
        assert not batchNorm, "the reference only instantiates batchNorm=False (train.py:255)"
        self.batchNorm, self.div_flow, self.full_res = batchNorm, div_flow, full_res
        self.conv1 = _conv(3, 64, 7, 2)
        self.conv2 = _conv(64, 128, 5, 2)
        self.conv3 = _conv(128, 256, 5, 2)
        self.conv_redir = _conv(256, 32, 1, 1)
        self.corr = correlate
        self.conv3_1 = _conv(473, 256)
        self.conv4, self.conv4_1 = _conv(256, 512, stride=2), _conv(512, 512)
        self.conv5, self.conv5_1 = _conv(512, 512, stride=2), _conv(512, 512)
        self.conv6, self.conv6_1 = _conv(512, 1024, stride=2), _conv(1024, 1024)
        self.deconv5, self.deconv4, self.deconv3 = _deconv(1024, 512), _deconv(1026, 256), _deconv(770, 128)
        self.deconv2, self.deconv1 = _deconv(386, 64), _deconv(194, 32)
        for lvl, cin in zip(range(6, 0, -1), (1024, 1026, 770, 386, 194, 98)):
            setattr(self, "predict_flow%d" % lvl, L.Conv2d(cin, 2, 3, 1, 1))
        for a, b in ((6, 5), (5, 4), (4, 3), (3, 2), (2, 1)):
            setattr(self, "upsampled_flow%d_to_%d" % (a, b), L.ConvTranspose2d(2, 2, 4, 2, 1))
        self.upsample1 = nn.Upsample(scale_factor=2, mode='bilinear')      # parameter-free; kept for attribute parity

    def init_weights(self):
        """FlowNetC6.py:89-101: U(0,1) bias first, then xavier weight."""
        for m in self.modules():
            if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
                if m.bias is not None:
                    nn.init.uniform_(m.bias)
                nn.init.xavier_uniform_(m.weight)

    def forward(self, x1, x2):
        c1a = self.conv1(x1)
        c2a = self.conv2(c1a)
        c3a = self.conv3(c2a)
        c3b = self.conv3(self.conv2(self.conv1(x2)))
        corr = F.leaky_relu(self.corr(c3a, c3b), _S)                        # corr_activation
        c31 = self.conv3_1(torch.cat((self.conv_redir(c3a), corr), 1))
        c4 = self.conv4_1(self.conv4(c31))
        c5 = self.conv5_1(self.conv5(c4))
        c6 = self.conv6_1(self.conv6(c5))
        flows = {6: self.predict_flow6(c6)}
        feat = c6
        for lvl, skip in ((5, c5), (4, c4), (3, c31), (2, c2a), (1, c1a)):
            dec = getattr(self, "deconv%d" % lvl)(feat)
            up = getattr(self, "upsampled_flow%d_to_%d" % (lvl + 1, lvl))(flows[lvl + 1])
            feat = torch.cat((skip, dec, up), 1)
            flows[lvl] = getattr(self, "predict_flow%d" % lvl)(feat)
        out = [flows[l] for l in range(1, 7)]
        if self.full_res:
            out = [self.div_flow * ops.upsample_bilinear2x(f) for f in out]
        if self.training:
            return tuple(out)
        return out[0]
