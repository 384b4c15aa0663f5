"""DeformConv / DeformConvPack (DCNv1) — class names, constructor arguments and parameter names of the reference's
modules/deform_conv.py:14-110."""
from ..functions.deform_conv_func import DeformConvFunction
from ._base import DeformConvBase


class DeformConv(DeformConvBase):

    def __init__(self, in_channels, out_channels, kernel_size, stride, padding, dilation=1, groups=1, deformable_groups=1,
                 im2col_step=64, bias=True):
        assert bias == False  # noqa: E712  (the reference asserts this too, deform_conv.py:20)
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, deformable_groups, im2col_step, bias)

    def forward(self, input, offset):
        assert 2 * self._taps() == offset.shape[1]
        return DeformConvFunction.apply(input, offset, *self._conv_args())


_DeformConv = DeformConvFunction.apply


class DeformConvPack(DeformConv):
    """Predicts its own offsets with a zero-initialised conv (`conv_offset`)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride, padding, dilation=1, groups=1, deformable_groups=1,
                 im2col_step=64, bias=True, lr_mult=0.1):
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, deformable_groups, im2col_step, bias)
        self.conv_offset = self._predictor(2, lr_mult)

    def init_offset(self):
        self.conv_offset.weight.data.zero_()
        self.conv_offset.bias.data.zero_()

    def forward(self, input):
        return DeformConvFunction.apply(input, self.conv_offset(input), *self._conv_args())
