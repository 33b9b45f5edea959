"""Layer set of the engine's networks.  Parameters/buffers are registered exactly like torch.nn's so
``state_dict()`` keys and shapes equal the reference's; the compute of every layer
goes through ``cc_amd.ops`` (hand-written gfx950 kernels behind the C ABI)."""
import torch
import torch.nn as nn

from . import ops


class Conv2d(nn.Conv2d):
    """nn.Conv2d parameters; forward = implicit-GEMM MFMA kernel with the activation fused into the epilogue.
    act: None | 'relu' | 'lrelu' (slope 0.2) | 'sigmoid'."""

    def __init__(self, cin, cout, kernel_size, stride=1, padding=0, bias=True, act=None, slope=0.0):
        super().__init__(cin, cout, kernel_size, stride, padding, bias=bias)
        self.act, self.slope = act, slope            # slope: LeakyReLU negative slope (0 -> 0.2)

    def forward(self, x):
        return ops.conv2d(x, self.weight, self.bias, self.stride[0], self.padding[0], self.act, None, 1.0, self.slope)


class ConvTranspose2d(nn.ConvTranspose2d):
    def __init__(self, cin, cout, kernel_size, stride=1, padding=0, output_padding=0, act=None, slope=0.0):
        super().__init__(cin, cout, kernel_size, stride, padding, output_padding)
        self.act, self.slope = act, slope

    def forward(self, x):
        return ops.conv_transpose2d(x, self.weight, self.bias, self.stride[0], self.padding[0],
                                    self.output_padding[0], self.act, self.slope)


class BatchNorm2d(nn.BatchNorm2d):
    def forward(self, x):
        return ops.batch_norm(x, self.weight, self.bias, self.running_mean, self.running_var,
                              self.num_batches_tracked, self.training, self.momentum, self.eps)


class Act(nn.Module):
    """Parameter-free placeholder keeping the reference's nn.Sequential indices (conv at .0, .2, ...);
    the activation itself is fused into the preceding convolution."""

    def forward(self, x):
        return x
