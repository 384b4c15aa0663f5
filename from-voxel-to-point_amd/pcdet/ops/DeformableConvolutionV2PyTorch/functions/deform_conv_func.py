"""DeformConvFunction (DCNv1) — call surface of the reference's functions/deform_conv_func.py:
apply(input, offset, weight, bias, stride, padding, dilation, group, deformable_groups, im2col_step)."""
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import DCN
from ._geometry import conv_geometry


class DeformConvFunction(Function):
    @staticmethod
    def forward(ctx, input, offset, weight, bias, *loose):
        ctx.geometry = conv_geometry(weight, *loose)
        ctx.save_for_backward(input, offset, weight, bias)
        return DCN.deform_conv_forward(input, weight, bias, offset, *ctx.geometry)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        input, offset, weight, bias = ctx.saved_tensors
        d_input, d_offset, d_weight, d_bias = DCN.deform_conv_backward(input, weight, bias, offset, grad_output.contiguous(), *ctx.geometry)
        return (d_input, d_offset, d_weight, d_bias) + (None,) * 6
