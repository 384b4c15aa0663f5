"""autograd Functions of the sparse-conv path — names and call signatures of reference
spconv/functional.py:20-175.  `indice_pairs` may be the reference-format pair tensor or a
`pcdet.ops.spconv.ops.Rulebook` (the modules in conv.py pass the latter, skipping pair materialisation)."""
import torch
from torch.autograd import Function

from . import ops as ops


def _is_tensor(x):
    return isinstance(x, torch.Tensor)


class _ConvBase(Function):
    INVERSE, SUBM = False, False

    @classmethod
    def _fwd(cls, ctx, features, filters, indice_pairs, indice_pair_num, num_activate_out):
        ctx.rulebook = ops._rulebook_of(indice_pairs, indice_pair_num, features.shape[0], num_activate_out, cls.INVERSE)
        ctx.save_for_backward(features, filters)
        return ops.indice_conv(features, filters, ctx.rulebook, indice_pair_num, num_activate_out, cls.INVERSE, cls.SUBM)

    @classmethod
    def _bwd(cls, ctx, grad_output):
        features, filters = ctx.saved_tensors
        input_bp, filters_bp = ops.indice_conv_backward(features, filters, grad_output.contiguous(), ctx.rulebook, None,
                                                        cls.INVERSE, cls.SUBM)
        return input_bp, filters_bp, None, None, None


class SparseConvFunction(_ConvBase):
    INVERSE, SUBM = False, False

    @staticmethod
    def forward(ctx, features, filters, indice_pairs, indice_pair_num, num_activate_out):
        return SparseConvFunction._fwd(ctx, features, filters, indice_pairs, indice_pair_num, num_activate_out)

    @staticmethod
    def backward(ctx, grad_output):
        return SparseConvFunction._bwd(ctx, grad_output)


class SparseInverseConvFunction(_ConvBase):
    INVERSE, SUBM = True, False

    @staticmethod
    def forward(ctx, features, filters, indice_pairs, indice_pair_num, num_activate_out):
        return SparseInverseConvFunction._fwd(ctx, features, filters, indice_pairs, indice_pair_num, num_activate_out)

    @staticmethod
    def backward(ctx, grad_output):
        return SparseInverseConvFunction._bwd(ctx, grad_output)


class SubMConvFunction(_ConvBase):
    INVERSE, SUBM = False, True

    @staticmethod
    def forward(ctx, features, filters, indice_pairs, indice_pair_num, num_activate_out):
        return SubMConvFunction._fwd(ctx, features, filters, indice_pairs, indice_pair_num, num_activate_out)

    @staticmethod
    def backward(ctx, grad_output):
        return SubMConvFunction._bwd(ctx, grad_output)


class SparseMaxPoolFunction(Function):

    @staticmethod
    def forward(ctx, features, indice_pairs, indice_pair_num, num_activate_out):
        ctx.rulebook = ops._rulebook_of(indice_pairs, indice_pair_num, features.shape[0], num_activate_out, False)
        out = ops.indice_maxpool(features, ctx.rulebook, indice_pair_num, num_activate_out)
        ctx.save_for_backward(features, out)
        return out

    @staticmethod
    def backward(ctx, grad_output):
        features, out = ctx.saved_tensors
        input_bp = ops.indice_maxpool_backward(features, out, grad_output.contiguous(), ctx.rulebook, None)
        return input_bp, None, None, None


class SparseGroupFunction(Function):
    """features (N_in, C) -> (K, N_out, C): neighbour features per kernel offset, zeros where absent."""
    SUBM = False

    @staticmethod
    def forward(ctx, features, indice_pairs, indice_pair_num, num_activate_out):
        ctx.rulebook = ops._rulebook_of(indice_pairs, indice_pair_num, features.shape[0], num_activate_out, False)
        ctx.save_for_backward(features)
        return ops.indice_group(features, ctx.rulebook, indice_pair_num, num_activate_out, False, False)

    @staticmethod
    def backward(ctx, grad_output):
        (features,) = ctx.saved_tensors
        return ops.indice_group_backward(features, grad_output.contiguous(), ctx.rulebook, None, False, False), None, None, None


class SubMGroupFunction(Function):
    SUBM = True

    @staticmethod
    def forward(ctx, features, indice_pairs, indice_pair_num, num_activate_out):
        ctx.rulebook = ops._rulebook_of(indice_pairs, indice_pair_num, features.shape[0], num_activate_out, False)
        ctx.save_for_backward(features)
        return ops.indice_group(features, ctx.rulebook, indice_pair_num, num_activate_out, False, True)

    @staticmethod
    def backward(ctx, grad_output):
        (features,) = ctx.saved_tensors
        return ops.indice_group_backward(features, grad_output.contiguous(), ctx.rulebook, None, False, True), None, None, None


indice_conv = SparseConvFunction.apply
indice_inverse_conv = SparseInverseConvFunction.apply
indice_subm_conv = SubMConvFunction.apply
indice_maxpool = SparseMaxPoolFunction.apply
indice_group = SparseGroupFunction.apply
indice_subm_group = SubMGroupFunction.apply
