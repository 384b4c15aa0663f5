"""ModulatedDeformConvFunction (DCNv2) — call surface of the reference's functions/modulated_deform_conv_func.py:15-56:
apply(input, offset, mask, weight, bias, stride, padding, dilation, groups, deformable_groups, im2col_step)."""
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import DCN
from ._geometry import conv_geometry


class ModulatedDeformConvFunction(Function):
    @staticmethod
    def forward(ctx, input, offset, mask, weight, bias, *loose):
        ctx.geometry = conv_geometry(weight, *loose)
        ctx.save_for_backward(input, offset, mask, weight, bias)
        return DCN.modulated_deform_conv_forward(input, weight, bias, offset, mask, *ctx.geometry)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        input, offset, mask, weight, bias = ctx.saved_tensors
        d_input, d_offset, d_mask, d_weight, d_bias = DCN.modulated_deform_conv_backward(
            input, weight, bias, offset, mask, grad_output, *ctx.geometry)
        return (d_input, d_offset, d_mask, d_weight, d_bias) + (None,) * 6
