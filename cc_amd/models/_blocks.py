"""Building blocks shared by the engine's networks (see cc_amd/nn.py for the layer set)."""
import torch
import torch.nn as nn

from .. import nn as L
from .. import ops


def xavier_zero_bias(net):
    """init_weights() of DispResNet6.py:138-143 / PoseNetB6.py:43-48 / MaskNet6.py:54-59."""
    for m in net.modules():
        if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
            nn.init.xavier_uniform_(m.weight.data)
            if m.bias is not None:
                m.bias.data.zero_()


def crop_like(t, ref):
    assert t.size(2) >= ref.size(2) and t.size(3) >= ref.size(3)
    if t.size(2) == ref.size(2) and t.size(3) == ref.size(3):
        return t
    return t[:, :, :ref.size(2), :ref.size(3)]


def seq_conv_act(cin, cout, k, stride, act):
    """nn.Sequential(conv, activation): conv parameters at index 0, activation fused into the conv epilogue."""
    return nn.Sequential(L.Conv2d(cin, cout, k, stride, (k - 1) // 2, act=act), L.Act())


def conv_call(m, x, pre_act=None, defer=False, residual=None):
    """L.Conv2d module `m` applied to x as a link of a conv -> conv chain: defer = every consumer of the output is a
    convolution called with pre_act=m.act (its data-gradient epilogue then applies this layer's activation backward)."""
    return ops.conv2d(x, m.weight, m.bias, m.stride[0], m.padding[0], m.act, residual, 1.0, m.slope, pre_act=pre_act, defer=defer)


class BasicBlock(nn.Module):
    """models/DispResNet6.py:14-43: conv3x3 -> ReLU -> conv3x3 -> (+ shortcut) -> ReLU, no BN on the main path.
    The residual add and the final ReLU run in conv2's epilogue."""

    def __init__(self, cin, cout, stride, downsample):
        super().__init__()
        self.conv1 = L.Conv2d(cin, cout, 3, stride, 1, bias=False, act="relu")
        self.conv2 = L.Conv2d(cout, cout, 3, 1, 1, bias=False, act="relu")
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        r = x if self.downsample is None else self.downsample(x)
        y = conv_call(self.conv1, x, defer=True)                    # consumed by conv2 only
        return ops.conv2d(y, self.conv2.weight, None, 1, 1, "relu", residual=r, pre_act="relu")


def basic_block(tape, blk, x, out=None):
    """BasicBlock on the tape (cc_amd/tape.py).  conv1 is recorded as the FIRST consumer of x: its data-gradient is then the
    last contribution to x's gradient and carries the shortcut's gradient and relu'(x) in its epilogue."""
    y = tape.conv(x, blk.conv1.weight, None, blk.stride, 1, "relu")
    if blk.downsample is None:
        r = x
    else:
        ds = blk.downsample
        r = tape.batch_norm(tape.conv(x, ds[0].weight, None, ds[0].stride[0], 0, None), ds[1])
    return tape.conv(y, blk.conv2.weight, None, 1, 1, "relu", residual=r, out=out)


def make_layer(cin, cout, blocks, stride):
    """models/DispResNet6.py:45-60: the first block gets a 1x1(stride)+BatchNorm shortcut when shapes change."""
    ds = None
    if stride != 1 or cin != cout:
        ds = nn.Sequential(L.Conv2d(cin, cout, 1, stride, 0, bias=False), L.BatchNorm2d(cout))
    layers = [BasicBlock(cin, cout, stride, ds)] + [BasicBlock(cout, cout, 1, None) for _ in range(1, blocks)]
    return nn.Sequential(*layers)
