"""Sparse convolution modules — class names, constructor arguments, parameter layout
(`weight [*kernel_size, Cin, Cout]`, optional `bias [Cout]`) and forward contract of the reference's
spconv/conv.py:48-480, running on the hashed rulebook + fused MFMA conv of libfv2p_ops.

mmcv is not required: if it is importable the classes are registered in its CONV_LAYERS registry as the
reference does (conv.py:17,233), otherwise registration is skipped."""
import math
import os

import numpy as np
import torch
from torch.nn import init
from torch.nn.parameter import Parameter

from . import functional as Fsp
from . import ops
from .modules import SparseModule
from .prefetch import geometry_matches
from .structure import SparseConvTensor

try:  # optional, registry only (no arithmetic lives in mmcv)
    from mmcv.cnn import CONV_LAYERS as _REG
    _register = _REG.register_module()
except Exception:  # noqa: BLE001
    def _register(cls):
        return cls


def _calculate_fan_in_and_fan_out_hwio(tensor):
    if tensor.ndimension() < 2:
        raise ValueError('fan in and fan out can not be computed for tensor with fewer than 2 dimensions')
    if tensor.ndimension() == 2:
        return tensor.size(-2), tensor.size(-1)
    rf = tensor[..., 0, 0].numel()
    return tensor.size(-2) * rf, tensor.size(-1) * rf


class SparseConvolution(SparseModule):

    def __init__(self, ndim, in_channels, out_channels, kernel_size=3, stride=1, padding=0, dilation=1, groups=1,
                 bias=True, subm=False, output_padding=0, transposed=False, inverse=False, indice_key=None,
                 fused_bn=False):
        super(SparseConvolution, self).__init__()
        assert groups == 1
        as_list = lambda v: list(v) if isinstance(v, (list, tuple)) else [v] * ndim
        kernel_size, stride, padding = as_list(kernel_size), as_list(stride), as_list(padding)
        dilation, output_padding = as_list(dilation), as_list(output_padding)
        for d, s in zip(dilation, stride):
            assert any([s == 1, d == 1]), "don't support this."
        self.ndim = ndim
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.kernel_size = kernel_size
        self.conv1x1 = np.prod(kernel_size) == 1
        self.stride = stride
        self.padding = padding
        self.dilation = dilation
        self.transposed = transposed
        self.inverse = inverse
        self.output_padding = output_padding
        self.groups = groups
        self.subm = subm
        self.indice_key = indice_key
        self.fused_bn = fused_bn
        self.weight = Parameter(torch.Tensor(*kernel_size, in_channels, out_channels))
        if bias:
            self.bias = Parameter(torch.Tensor(out_channels))
        else:
            self.register_parameter('bias', None)
        self.reset_parameters()

    def reset_parameters(self):
        init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            fan_in, _ = _calculate_fan_in_and_fan_out_hwio(self.weight)
            bound = 1 / math.sqrt(fan_in)
            init.uniform_(self.bias, -bound, bound)

    def forward(self, input, _post=None):
        """`_post` (internal, set by SparseSequential): the (BatchNorm1d, ReLU-or-None) pair that follows this conv; when
        the compiled binding can run conv -> BN -> ReLU in one call the returned tensor carries `_fv2p_post_done`."""
        assert isinstance(input, SparseConvTensor)
        if _post is not None and not self.fused_bn and FOLD_SEQUENTIAL:
            folded = conv_bn_fold(self, input, _post[0], _post[1])
            if folded is not None:
                folded._fv2p_post_done = True
                return folded
        materialise_pending(input)   # (a BatchNorm left pending by conv_bn_fold(defer=True): this path reads materialised rows)
        # alias of self.weight made by defer_weight_gradients() for this pass (its dW is joined at the end of backward)
        weight = self.__dict__.pop("_fv2p_gated_weight", None)
        if weight is None or not torch.is_grad_enabled():
            weight = self.weight
        features = input.features
        indices = input.indices
        spatial_shape = input.spatial_shape
        batch_size = input.batch_size
        if self.subm:
            out_spatial_shape = spatial_shape
        elif self.transposed:
            out_spatial_shape = ops.get_deconv_output_size(spatial_shape, self.kernel_size, self.stride, self.padding,
                                                           self.dilation, self.output_padding)
        else:
            out_spatial_shape = ops.get_conv_output_size(spatial_shape, self.kernel_size, self.stride, self.padding,
                                                         self.dilation)
        if self.conv1x1:  # reference conv.py:137-148: plain GEMM on the feature matrix
            features = torch.mm(input.features, weight.view(self.in_channels, self.out_channels))
            if self.bias is not None:
                features += self.bias
            out_tensor = SparseConvTensor(features, input.indices, input.spatial_shape, input.batch_size)
            out_tensor.indice_dict = input.indice_dict
            out_tensor.grid = input.grid
            return out_tensor
        datas = input.find_indice_pair(self.indice_key)
        if self.inverse:
            assert datas is not None and self.indice_key is not None
            rb = datas
            outids, out_spatial_shape = rb.indices, rb.spatial_shape
            assert rb.kvol == np.prod(self.kernel_size), 'inverse conv must have same kernel size as its couple conv'
        else:
            if self.indice_key is not None and datas is not None and geometry_matches(datas, self, indices):
                rb = datas
            else:
                rb = ops.build_rulebook(indices, batch_size, spatial_shape, self.kernel_size, self.stride, self.padding,
                                        self.dilation, self.output_padding, self.subm, self.transposed)
                input.indice_dict[self.indice_key] = rb
            outids = rb.outids
        n_out = outids.shape[0]
        if _post is not None and not self.fused_bn:
            fused = self._conv_bn_relu(features, rb, n_out, _post, weight)
            if fused is not None:
                out_tensor = SparseConvTensor(fused, outids, out_spatial_shape, batch_size)
                out_tensor.indice_dict = input.indice_dict
                out_tensor.grid = input.grid
                out_tensor._fv2p_post_done = True
                return out_tensor
        if self.fused_bn:
            assert self.bias is not None
            out_features = ops.fused_indice_conv(features, weight, self.bias, rb, None, n_out,
                                                 self.inverse, self.subm)
        else:
            if self.subm:
                out_features = Fsp.indice_subm_conv(features, weight, rb, None, n_out)
            elif self.inverse:
                out_features = Fsp.indice_inverse_conv(features, weight, rb, None, n_out)
            else:
                out_features = Fsp.indice_conv(features, weight, rb, None, n_out)
            if self.bias is not None:
                out_features += self.bias
        out_tensor = SparseConvTensor(out_features, outids, out_spatial_shape, batch_size)
        out_tensor.indice_dict = input.indice_dict
        out_tensor.grid = input.grid
        return out_tensor


def _conv_bn_relu(self, features, rb, n_out, post, weight=None):
    """conv -> BatchNorm1d (-> ReLU) through the compiled binding in one call, or None when any of the three is not the
    plain case (then the modules run one by one, with torch's own error behaviour)."""
    import fv2p_native as _nat
    from .norm import fusable
    ext = _nat.torch_ext()
    bn, relu = post
    weight = self.weight if weight is None else weight
    if ext is None or n_out < 2 or not fusable(bn, relu, features, self.out_channels):
        return None
    if not (features.is_cuda and features.dtype == torch.float32 and self.weight.dtype == torch.float32 and features.dim() == 2):
        return None
    cin, cout = self.in_channels, self.out_channels   # forward gathers cin-channel rows, backward-data cout-channel rows
    (tab_f, flip_f), (tab_b, flip_b) = (rb.in_table(cin), rb.out_table(cout)) if self.inverse else (rb.out_table(cin), rb.in_table(cout))
    centre = (rb.kvol // 2) if (rb.subm and rb.tab_out is None and not self.inverse) else -1
    wpairs, wnum = rb.pairs_for_wgrad(cin, cout)
    out = ext.sparse_conv_bn_relu(features, weight, tab_f, flip_f, tab_b, flip_b, n_out, centre, wpairs,
                                  wnum, 1 if self.inverse else 0, self.bias, bn.weight, bn.bias, bn.running_mean,
                                  bn.running_var, bn.num_batches_tracked, bn.training,
                                  -1.0 if bn.momentum is None else float(bn.momentum), float(bn.eps), relu is not None,
                                  None if self.inverse else getattr(rb, "_perm_in", None))
    return out


SparseConvolution._conv_bn_relu = _conv_bn_relu


# conv -> BatchNorm1d -> ReLU inside a SparseSequential (post_act_block) through conv_bn_fold as well?  Measured (plain VoxelBackBone8x,
# batch 4, profiles/r06_*): NO - a conv that finalises its statistics itself ends 4.6 us later (its waves wait for their fp64 atomics,
# the last workgroup folds and clears the slots), the apply launch it feeds is only 2.2 us shorter (5.4 against 7.6 us): 1.79 against
# 1.68 ms per step.  The residual blocks are where the arrangement pays (no bias add, no separate reduce pass, no add / ReLU launches, no
# relu(bn1(y1)) tensor: VoxelResBackBone8x 4.01 against 4.47 ms per step), so only they use it (fv2p_harness/backbone.py); the flag keeps
# the path testable (tests/test_bn_fold_gpu.py).
FOLD_SEQUENTIAL = os.environ.get("FV2P_FOLD_SEQUENTIAL", "0") == "1"


def fold_enabled():
    """Round-6 arrangement of conv / BatchNorm / residual blocks (csrc_torch/fv2p_torch.cpp: conv_fin, bn_apply): on unless FV2P_BN_FOLD=0
    or `set_bn_fold(False)`; needs the compiled binding."""
    import fv2p_native as _nat
    ext = _nat.torch_ext()
    return ext is not None and ext.bn_fold()


def set_bn_fold(on):
    import fv2p_native as _nat
    ext = _nat.torch_ext()
    if ext is not None:
        ext.set_bn_fold(bool(on))


def materialise_pending(x):
    """A SparseConvTensor whose features are still the raw conv output with a BatchNorm (+ReLU) pending (conv_bn_fold(defer=True)):
    apply it now (one launch, statistics already final) - for consumers that cannot normalise on their gather."""
    pend = x.__dict__.pop("_fv2p_pending", None)
    if pend is not None:
        import fv2p_native as _nat
        saved, bn, relu, batch_stats = pend
        x.features = _nat.torch_ext().bn_apply(x.features, saved, bn.weight, bn.bias, relu, None, batch_stats)
    return x


def conv_bn_fold(conv, x, bn, relu_module, defer=False, residual=None):
    """conv -> BatchNorm1d (-> ReLU) of the reference's blocks (spconv_backbone.py:8-27, 32-68) on the round-6 kernels:

      * the conv's last workgroup finalises the BatchNorm statistics (mean / invstd / running statistics), so the normalisation is ONE
        launch that folds nothing - or none at all:
      * defer=True returns the RAW conv output with the BatchNorm (+ReLU) pending on the tensor; a following conv_bn_fold call on a
        conv whose kernel can normalise its gathered source rows (conv_fin's `pre_*`; ext.prenorm_supported) never materialises
        relu(bn(y)) - the bn1 -> relu -> conv2 of a residual block.  Any other consumer calls materialise_pending();
      * residual: out = relu?(bn(conv(x)) + residual) in the normalisation's launch (the `out.features += identity; relu` tail).

    Returns the output SparseConvTensor, or None when the situation is not the plain one (hooks, CPU, autocast, a conv bias without a
    train-mode BatchNorm behind it ...): the caller then runs the modules one by one, with torch's own behaviour."""
    import fv2p_native as _nat
    from .norm import fusable
    ext = _nat.torch_ext()
    if ext is None or not ext.bn_fold() or not isinstance(conv, SparseConvolution) or conv.conv1x1 or conv.fused_bn or conv.inverse:
        return None   # (not a SparseConvolution: e.g. the host mirror the parity tests put in a block's place)
    if conv._forward_hooks or conv._forward_pre_hooks or conv._backward_hooks:
        return None
    features = x.features
    if not (torch.is_tensor(features) and features.is_cuda and features.dtype == torch.float32 and features.dim() == 2
            and conv.weight.dtype == torch.float32 and x.indices.shape[0] != 0):
        return None
    if not fusable(bn, relu_module, features, conv.out_channels):
        return None
    weight = conv.__dict__.get("_fv2p_gated_weight")   # alias made by defer_weight_gradients() for this pass (popped once it is used)
    if weight is None or not torch.is_grad_enabled():
        weight = conv.weight
    indices, spatial_shape, batch_size = x.indices, x.spatial_shape, x.batch_size
    if conv.subm:
        out_spatial_shape = spatial_shape
    elif conv.transposed:
        out_spatial_shape = ops.get_deconv_output_size(spatial_shape, conv.kernel_size, conv.stride, conv.padding, conv.dilation, conv.output_padding)
    else:
        out_spatial_shape = ops.get_conv_output_size(spatial_shape, conv.kernel_size, conv.stride, conv.padding, conv.dilation)
    datas = x.find_indice_pair(conv.indice_key)
    if conv.indice_key is not None and datas is not None and geometry_matches(datas, conv, indices):
        rb = datas
    else:
        rb = ops.build_rulebook(indices, batch_size, spatial_shape, conv.kernel_size, conv.stride, conv.padding, conv.dilation,
                                conv.output_padding, conv.subm, conv.transposed)
        x.indice_dict[conv.indice_key] = rb
    outids = rb.outids
    n_out = outids.shape[0]
    if n_out < 2:
        return None
    cin, cout = conv.in_channels, conv.out_channels
    (tab_f, flip_f), (tab_b, flip_b) = rb.out_table(cin), rb.in_table(cout)
    centre = (rb.kvol // 2) if (rb.subm and rb.tab_out is None) else -1
    wpairs, wnum = rb.pairs_for_wgrad(cin, cout)
    pend = x.__dict__.get("_fv2p_pending")
    pre = (None, None, None, False, False)
    if pend is not None:
        if ext.prenorm_supported(cin, cout, rb.kvol, n_out, int(flip_f)):
            saved_src, bn_src, relu_src, bs_src = pend
            pre = (saved_src, bn_src.weight, bn_src.bias, relu_src, bs_src)
        else:
            features = materialise_pending(x).features
    momentum = -1.0 if bn.momentum is None else float(bn.momentum)
    res = ext.conv_fin(features, weight, tab_f, flip_f, tab_b, flip_b, n_out, centre, wpairs, wnum, 0,
                       getattr(rb, "_perm_in", None), conv.bias, True, bn.running_mean, bn.running_var, bn.num_batches_tracked, bn.training,
                       momentum, float(bn.eps), *pre)
    if not res:
        if pend is not None:
            materialise_pending(x)
        return None
    conv.__dict__.pop("_fv2p_gated_weight", None)
    y, saved = res
    batch_stats = bool(bn.training or bn.running_mean is None)
    out = SparseConvTensor(y, outids, out_spatial_shape, batch_size)
    out.indice_dict = x.indice_dict
    out.grid = x.grid
    if defer and residual is None:
        out._fv2p_pending = (saved, bn, relu_module is not None, batch_stats)
    else:
        out.features = ext.bn_apply(y, saved, bn.weight, bn.bias, relu_module is not None, residual, batch_stats)
    return out


def defer_weight_gradients(module):
    """Call at the top of a model's forward pass (training): the weight gradients of all sparse convolutions under
    `module` are then joined from their side stream once, at the end of the backward pass, instead of at the end of every
    conv's backward (csrc_torch/fv2p_torch.cpp, WeightGateFn) — the training stream no longer idles behind each dW.
    Gradients, hooks and DistributedDataParallel see the same values; conv weights just become ready last.  No-op
    without the compiled binding, under no_grad, or for CPU / frozen weights."""
    import fv2p_native as _nat
    ext = _nat.torch_ext()
    if ext is None or not torch.is_grad_enabled() or os.environ.get("FV2P_DEFER_WGRAD", "1") == "0":
        return
    convs = module.__dict__.get("_fv2p_convs")
    if convs is None:
        convs = module.__dict__["_fv2p_convs"] = [m for m in module.modules() if isinstance(m, SparseConvolution)]
    live = [m for m in convs if m.weight.requires_grad and m.weight.is_cuda and m.weight.dtype == torch.float32]
    if live:
        for m, w in zip(live, ext.gate_weights([m.weight for m in live])):
            m.__dict__["_fv2p_gated_weight"] = w


def _make(name, ndim, **fixed):
    """Builds the thin reference subclasses (conv.py:233-480)."""
    takes_geometry = not fixed.get("inverse", False)

    if takes_geometry:
        def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                     indice_key=None):
            SparseConvolution.__init__(self, ndim, in_channels, out_channels, kernel_size, stride, padding, dilation,
                                       groups, bias, indice_key=indice_key, **fixed)
    else:
        def __init__(self, in_channels, out_channels, kernel_size, indice_key, bias=True):
            SparseConvolution.__init__(self, ndim, in_channels, out_channels, kernel_size, bias=bias,
                                       indice_key=indice_key, **fixed)
    cls = type(name, (SparseConvolution,), {"__init__": __init__, "__module__": __name__})
    return _register(cls)


SparseConv2d = _make("SparseConv2d", 2)
SparseConv3d = _make("SparseConv3d", 3)
SparseConv4d = _make("SparseConv4d", 4)
SparseConvTranspose2d = _make("SparseConvTranspose2d", 2, transposed=True)
SparseConvTranspose3d = _make("SparseConvTranspose3d", 3, transposed=True)
SparseInverseConv2d = _make("SparseInverseConv2d", 2, inverse=True)
SparseInverseConv3d = _make("SparseInverseConv3d", 3, inverse=True)
SubMConv2d = _make("SubMConv2d", 2, subm=True)
SubMConv3d = _make("SubMConv3d", 3, subm=True)
SubMConv4d = _make("SubMConv4d", 4, subm=True)
