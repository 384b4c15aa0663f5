"""ModulatedDeformConv / ModulatedDeformConvPack (DCNv2) — class names, constructor arguments and parameter names of
the reference's modules/modulated_deform_conv.py:14-103."""
import torch

from ..functions.modulated_deform_conv_func import ModulatedDeformConvFunction
from ._base import DeformConvBase


class ModulatedDeformConv(DeformConvBase):

    def __init__(self, in_channels, out_channels, kernel_size, stride, padding, dilation=1, groups=1, deformable_groups=1,
                 im2col_step=64, bias=True):
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, deformable_groups, im2col_step, bias)

    def forward(self, input, offset, mask):
        assert 2 * self._taps() == offset.shape[1]
        assert self._taps() == mask.shape[1]
        return ModulatedDeformConvFunction.apply(input, offset, mask, *self._conv_args())


_ModulatedDeformConv = ModulatedDeformConvFunction.apply


class ModulatedDeformConvPack(ModulatedDeformConv):
    """Predicts its own (offset_h, offset_w, mask) with a zero-initialised conv (`conv_offset_mask`)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride, padding, dilation=1, groups=1, deformable_groups=1,
                 im2col_step=64, bias=True, lr_mult=0.1):
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, deformable_groups, im2col_step, bias)
        self.conv_offset_mask = self._predictor(3, lr_mult)

    def init_offset(self):
        self.conv_offset_mask.weight.data.zero_()
        self.conv_offset_mask.bias.data.zero_()

    def forward(self, input):
        o1, o2, mask = torch.chunk(self.conv_offset_mask(input), 3, dim=1)
        return ModulatedDeformConvFunction.apply(input, torch.cat((o1, o2), dim=1), torch.sigmoid(mask), *self._conv_args())
