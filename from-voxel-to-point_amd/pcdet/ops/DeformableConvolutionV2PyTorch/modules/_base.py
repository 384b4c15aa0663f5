"""Shared constructor of the deformable conv modules: geometry attributes, `weight [Cout, Cin/groups, kh, kw]` and a
`bias [Cout]` parameter that exists even for bias=False (then frozen, but still added by the op — reference
modules/modulated_deform_conv.py:36-41, deform_conv.py:37-42)."""
import math

import torch
from torch import nn
from torch.nn import init
from torch.nn.modules.utils import _pair


class DeformConvBase(nn.Module):

    def __init__(self, in_channels, out_channels, kernel_size, stride, padding, dilation, groups, deformable_groups, im2col_step, bias):
        super().__init__()
        for name, ch in (("in_channels", in_channels), ("out_channels", out_channels)):
            if ch % groups != 0:
                raise ValueError('{} {} must be divisible by groups {}'.format(name, ch, groups))
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride = _pair(kernel_size), _pair(stride)
        self.padding, self.dilation = _pair(padding), _pair(dilation)
        self.groups, self.deformable_groups, self.im2col_step = groups, deformable_groups, im2col_step
        self.use_bias = bias
        self.weight = nn.Parameter(torch.Tensor(out_channels, in_channels // groups, *self.kernel_size))
        self.bias = nn.Parameter(torch.Tensor(out_channels))
        self.reset_parameters()
        if not self.use_bias:
            self.bias.requires_grad = False

    def reset_parameters(self):
        init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            fan_in, _ = init._calculate_fan_in_and_fan_out(self.weight)
            bound = 1 / math.sqrt(fan_in)
            init.uniform_(self.bias, -bound, bound)

    def _taps(self):
        return self.deformable_groups * self.kernel_size[0] * self.kernel_size[1]

    def _conv_args(self):
        return (self.weight, self.bias, self.stride, self.padding, self.dilation, self.groups, self.deformable_groups, self.im2col_step)

    def _predictor(self, per_tap, lr_mult):
        conv = nn.Conv2d(self.in_channels, self._taps() * per_tap, kernel_size=self.kernel_size, stride=self.stride,
                         padding=self.padding, bias=True)
        conv.lr_mult = lr_mult
        conv.weight.data.zero_()
        conv.bias.data.zero_()
        return conv
