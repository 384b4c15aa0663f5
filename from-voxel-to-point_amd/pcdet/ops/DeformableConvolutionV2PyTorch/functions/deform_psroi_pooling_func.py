"""DeformRoIPoolingFunction — call surface of the reference's functions/deform_psroi_pooling_func.py:15-64."""
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import DCN


class DeformRoIPoolingFunction(Function):
    @staticmethod
    def forward(ctx, input, rois, offset, spatial_scale, pooled_size, output_dim, no_trans, group_size=1, part_size=None,
                sample_per_part=4, trans_std=.0):
        ctx.conf = (int(no_trans), spatial_scale, output_dim, group_size, pooled_size, pooled_size if part_size is None else part_size,
                    sample_per_part, trans_std)
        output, output_count = DCN.deform_psroi_pooling_forward(input, rois, offset, *ctx.conf)
        ctx.save_for_backward(input, rois, offset, output_count)
        return output

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        input, rois, offset, output_count = ctx.saved_tensors
        grad_input, grad_offset = DCN.deform_psroi_pooling_backward(grad_output, input, rois, offset, output_count, *ctx.conf)
        return (grad_input, None, grad_offset) + (None,) * 8
