"""SparseModule / SparseSequential / ToDense / RemoveGrid — surface of reference spconv/modules.py:46-202."""
import sys
from collections import OrderedDict

import torch
from torch import nn

from .norm import batch_norm_relu
from .structure import SparseConvTensor


def is_spconv_module(module):
    return isinstance(module, (SparseModule,))


def is_sparse_conv(module):
    from .conv import SparseConvolution
    return isinstance(module, SparseConvolution)


class SparseModule(nn.Module):
    """Marker base class: subclasses receive the SparseConvTensor itself inside SparseSequential."""
    pass


class SparseSequential(SparseModule):
    """Sequential container: sparse modules get the tensor, plain nn.Modules are applied to `.features`
    in place (reference modules.py:125-136 — models rely on that in-place mutation)."""

    def __init__(self, *args, **kwargs):
        super(SparseSequential, self).__init__()
        if len(args) == 1 and isinstance(args[0], OrderedDict):
            for key, module in args[0].items():
                self.add_module(key, module)
        else:
            for idx, module in enumerate(args):
                self.add_module(str(idx), module)
        for name, module in kwargs.items():
            if sys.version_info < (3, 6):
                raise ValueError('kwargs only supported in py36+')
            if name in self._modules:
                raise ValueError('name exists.')
            self.add_module(name, module)
        self._sparity_dict = {}

    def __getitem__(self, idx):
        if not (-len(self) <= idx < len(self)):
            raise IndexError('index {} is out of range'.format(idx))
        if idx < 0:
            idx += len(self)
        return list(self._modules.values())[idx]

    def __len__(self):
        return len(self._modules)

    @property
    def sparity_dict(self):
        return self._sparity_dict

    def add(self, module, name=None):
        if name is None:
            name = str(len(self._modules))
            if name in self._modules:
                raise KeyError('name exists')
        self.add_module(name, module)

    def forward(self, input):
        items = list(self._modules.items())
        i = 0
        while i < len(items):
            k, module = items[i]
            i += 1
            if is_spconv_module(module):
                assert isinstance(input, SparseConvTensor)
                self._sparity_dict[k] = input.sparity
                hooked = bool(module._forward_hooks or module._forward_pre_hooks or module._backward_hooks)   # a hook on the conv must see the conv's own output
                if is_sparse_conv(module) and not hooked and i < len(items) and isinstance(items[i][1], nn.BatchNorm1d) and input.indices.shape[0] != 0:
                    # conv -> BatchNorm1d (-> ReLU), the reference's post_act_block: offered to the conv as one fused call
                    relu = items[i + 1][1] if i + 1 < len(items) and isinstance(items[i + 1][1], nn.ReLU) else None
                    input = module(input, _post=(items[i][1], relu))
                    if getattr(input, "_fv2p_post_done", False):
                        input._fv2p_post_done = False
                        i += 1 + (relu is not None)
                else:
                    input = module(input)
            elif isinstance(input, SparseConvTensor):
                if input.indices.shape[0] != 0:
                    if isinstance(module, nn.BatchNorm1d):
                        # BatchNorm1d (+ReLU) pair of the reference's post_act_block: one fused HIP op (norm.py)
                        nxt = items[i][1] if i < len(items) and isinstance(items[i][1], nn.ReLU) else None
                        fused = batch_norm_relu(module, input.features, nxt)
                        if fused is not None:
                            input.features = fused
                            i += nxt is not None
                            continue
                    input.features = module(input.features)
            else:
                input = module(input)
        return input

    def fused(self):
        """Folds every (SparseConvolution, BatchNorm1d) pair into one biased conv for inference
        (the reference's `fused()` is documented "don't use this", modules.py:138-185; the folding here
        uses the standard sqrt(var + eps) so the fused net equals conv -> eval-mode BN)."""
        mods = list(self._modules.values())
        out, i = [], 0
        while i < len(mods):
            m = mods[i]
            if is_sparse_conv(m) and i + 1 < len(mods) and isinstance(mods[i + 1], nn.BatchNorm1d):
                out.append(fold_conv_bn(m, mods[i + 1]))
                i += 2
            else:
                out.append(m)
                i += 1
        return SparseSequential(*out)


def fold_conv_bn(m, bn):
    """One biased SparseConvolution = conv `m` followed by BatchNorm1d `bn` in eval mode (inference):
    W' = W * gamma / sqrt(var + eps),  b' = (b - mean) * gamma / sqrt(var + eps) + beta."""
    from .conv import SparseConvolution
    conv = SparseConvolution(ndim=m.ndim, in_channels=m.in_channels, out_channels=m.out_channels,
                             kernel_size=m.kernel_size, stride=m.stride, padding=m.padding,
                             dilation=m.dilation, groups=m.groups, bias=True, subm=m.subm,
                             output_padding=m.output_padding, transposed=m.transposed, inverse=m.inverse,
                             indice_key=m.indice_key, fused_bn=True)
    conv.to(m.weight.device)
    with torch.no_grad():
        gamma = bn.weight if bn.weight is not None else torch.ones_like(bn.running_var)
        beta = bn.bias if bn.bias is not None else torch.zeros_like(bn.running_mean)
        scale = gamma / torch.sqrt(bn.running_var + bn.eps)
        conv.weight.copy_(m.weight * scale)
        b0 = m.bias if m.bias is not None else torch.zeros_like(bn.running_mean)
        conv.bias.copy_((b0 - bn.running_mean) * scale + beta)
    return conv


class ToDense(SparseModule):
    """SparseConvTensor -> dense N C (D) H W tensor."""

    def forward(self, x: SparseConvTensor):
        return x.dense()


class RemoveGrid(SparseModule):
    """Drops the (unused) pre-allocated grid buffer."""

    def forward(self, x: SparseConvTensor):
        x.grid = None
        return x
