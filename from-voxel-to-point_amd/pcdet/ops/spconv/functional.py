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


def _dispatch(fn_cls):
    """reference functional.py:169-171 binds `X.apply`; here the call first tries the compiled autograd Function
    (lib/fv2p_torch.so: same kernels, backward without Python) and otherwise runs the Python Function above."""
    import fv2p_native as _nat

    def call(features, filters, indice_pairs, indice_pair_num, num_activate_out):
        ext = _nat.torch_ext()
        if (ext is not None and features.is_cuda and features.dtype == torch.float32 and filters.dtype == torch.float32
                and features.dim() == 2 and not torch.is_autocast_enabled()):
            rb = ops._rulebook_of(indice_pairs, indice_pair_num, features.shape[0], num_activate_out, fn_cls.INVERSE)
            cin, cout = filters.shape[-2], filters.shape[-1]   # forward gathers cin-channel rows, backward-data cout-channel rows
            (tab_f, flip_f), (tab_b, flip_b) = (rb.in_table(cin), rb.out_table(cout)) if fn_cls.INVERSE else (rb.out_table(cin), rb.in_table(cout))
            centre = (rb.kvol // 2) if (rb.subm and rb.tab_out is None and not fn_cls.INVERSE) else -1
            wpairs, wnum = rb.pairs_for_wgrad(cin, cout)   # pair lists (prefetched, or built here for wide submanifold layers): pair-split wgrad
            return ext.sparse_conv(features, filters, tab_f, flip_f, tab_b, flip_b, num_activate_out, centre,
                                   wpairs, wnum, 1 if fn_cls.INVERSE else 0,
                                   None if fn_cls.INVERSE else getattr(rb, "_perm_in", None))
        return fn_cls.apply(features, filters, indice_pairs, indice_pair_num, num_activate_out)

    call.__name__ = fn_cls.__name__ + "_apply"
    return call


indice_conv = _dispatch(SparseConvFunction)
indice_inverse_conv = _dispatch(SparseInverseConvFunction)
indice_subm_conv = _dispatch(SubMConvFunction)
indice_maxpool = SparseMaxPoolFunction.apply
indice_group = SparseGroupFunction.apply
indice_subm_group = SubMGroupFunction.apply
