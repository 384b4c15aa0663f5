"""BatchNorm1d (+ReLU) on sparse-tensor features as two HIP launches forward and two backward.

The reference builds every backbone block as ``SparseSequential(conv, nn.BatchNorm1d(eps=1e-3, momentum=0.01),
nn.ReLU())`` (pcdet/models/backbones_3d/spconv_backbone.py:8-27, :75) and ``SparseSequential.forward`` applies the two
torch modules to ``.features`` (pcdet/ops/spconv/modules.py:86-100).  ``batch_norm_relu`` is what this package's
``SparseSequential`` calls for that pair instead: same parameters, same running-statistics update, same result within
fp32 rounding, kernels in csrc/batchnorm.hip.  It returns None whenever the situation is not the plain one (hooks on
the modules, autocast, non-fp32, CPU tensors, a single row in training mode ...) and the caller then runs the torch
modules one by one, so error behaviour stays torch's.
"""
import os

import torch
from torch import nn
from torch.autograd import Function

import fv2p_native as _nat

_ENABLED = os.environ.get("FV2P_FUSED_BN", "1") != "0"


_WS_BYTES = {}


def _ws_bytes(c):
    b = _WS_BYTES.get(c)
    if b is None:
        b = _WS_BYTES[c] = _nat.call("fv2p_batchnorm_ws_bytes", 0, c)
    return b


class _BatchNormReLU(Function):
    @staticmethod
    def forward(ctx, x, weight, bias, bn, relu):
        n, c = x.shape
        dev = x.device
        batch_stats = bn.training or bn.running_mean is None
        y = torch.empty_like(x)
        with _nat.device_guard(dev):
            if batch_stats:
                stats = torch.empty((2, c), dtype=torch.float32, device=dev)
                mean, invstd = stats[0], stats[1]
                track = bn.training and bn.running_mean is not None
                ws = _nat.workspace(_ws_bytes(c), dev)
                _nat.call("fv2p_batchnorm_forward", x, n, c, float(bn.eps), -1.0 if bn.momentum is None else float(bn.momentum),
                          weight, bias, int(relu), bn.running_mean if track else None, bn.running_var if track else None,
                          bn.num_batches_tracked if track else None, mean, invstd, y, ws, ws.numel(), _nat.stream())
            else:
                mean = bn.running_mean
                invstd = torch.rsqrt(bn.running_var + bn.eps)
                _nat.call("fv2p_batchnorm_apply", x, n, c, mean, invstd, weight, bias, int(relu), y, _nat.stream())
        ctx.save_for_backward(x, mean, invstd, weight, bias)
        ctx.relu, ctx.batch_stats = bool(relu), bool(batch_stats)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, mean, invstd, weight, bias = ctx.saved_tensors
        n, c = x.shape
        dev = x.device
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        dpar = torch.empty((2, c), dtype=torch.float32, device=dev)
        dgamma, dbeta = dpar[0], dpar[1]
        with _nat.device_guard(dev):
            ws = _nat.workspace(_ws_bytes(c), dev)
            _nat.call("fv2p_batchnorm_backward", x, dy, n, c, mean, invstd, weight, bias, int(ctx.relu), int(ctx.batch_stats),
                      dx, dgamma, dbeta, ws, ws.numel(), _nat.stream())
        return (dx if ctx.needs_input_grad[0] else None, dgamma if (weight is not None and ctx.needs_input_grad[1]) else None,
                dbeta if (bias is not None and ctx.needs_input_grad[2]) else None, None, None)


def _plain(module):
    return not (module._forward_hooks or module._forward_pre_hooks or module._backward_hooks)


def fusable(bn, relu_module, x, channels):
    """The plain case of the (BatchNorm1d, ReLU) pair on features x [N, channels] the fused op covers."""
    if not _ENABLED or type(bn) is not nn.BatchNorm1d or not _plain(bn):
        return False
    if relu_module is not None and (type(relu_module) is not nn.ReLU or not _plain(relu_module)):
        return False
    if not (torch.is_tensor(x) and x.is_cuda and x.dtype == torch.float32 and x.dim() == 2):
        return False
    if torch.is_autocast_enabled() or channels != bn.num_features:
        return False
    c = channels
    if c > 256 and c % 4 != 0 or c > 1024:
        return False
    if bn.weight is not None and (bn.weight.dtype != torch.float32 or not bn.weight.is_cuda):
        return False
    return True


def batch_norm_relu(bn, x, relu_module=None):
    """y = relu?(bn(x)) for x [N, C] fp32 on the GPU, or None when the fused path does not apply."""
    if not (torch.is_tensor(x) and x.dim() == 2 and x.is_contiguous() and fusable(bn, relu_module, x, x.shape[1])):
        return None
    if x.shape[0] < 2:
        return None
    ext = _nat.torch_ext()
    if ext is not None:
        return ext.batch_norm_relu(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked, bn.training,
                                   -1.0 if bn.momentum is None else float(bn.momentum), float(bn.eps), relu_module is not None)
    return _BatchNormReLU.apply(x, bn.weight, bn.bias, bn, relu_module is not None)
