"""Weight gradients of the dense 2-D convolutions (BEV backbone, heads) beside their data gradients.

torch's convolution_backward computes dX and dW of a Conv2d one after the other on the calling stream; in a detector's backward pass
the dense branch is the chain every other stream waits for, and half of its convolution time is dW, which nothing
downstream needs before the optimiser.  `gate(modules)` — called at the top of the branch's forward pass — routes the conv weights
through the library's weight gate (csrc_torch/fv2p_torch.cpp: WeightGateFn, the one the sparse convs use) and the patched forward of
the modules calls `dense_conv`: at::convolution with a backward that runs dW on the weight-gradient side stream, joined once by the
gate when the last conv of the branch has produced its gradient.  The kernels are MIOpen's, exactly those torch would launch; values,
hooks and DistributedDataParallel see the same gradients.  Harness scheduling, not part of pcdet.ops."""
import types

import torch
import torch.nn as nn

import fv2p_native as _nat


def _conv_forward(self, x):
    w = self.__dict__.get("_fv2p_gated_weight")
    ext = _nat.torch_ext()
    if w is None or ext is None or not x.is_cuda or not isinstance(self.padding, tuple) or self.padding_mode != "zeros":
        return type(self).forward(self, x)
    return ext.dense_conv(x, w, self.bias, list(self.stride), list(self.padding), list(self.dilation), False, [0] * len(self.stride), self.groups)


def _deconv_forward(self, x, output_size=None):
    w = self.__dict__.get("_fv2p_gated_weight")
    ext = _nat.torch_ext()
    if w is None or ext is None or not x.is_cuda or output_size is not None or self.padding_mode != "zeros":
        return type(self).forward(self, x, output_size)
    return ext.dense_conv(x, w, self.bias, list(self.stride), list(self.padding), list(self.dilation), True, list(self.output_padding), self.groups)


def _convs(modules):
    out = []
    for mod in modules:
        for m in mod.modules():
            if type(m) in (nn.Conv2d, nn.ConvTranspose2d):
                out.append(m)
    return out


def gate(modules, on=True):
    """Call at the top of the dense branch's forward pass.  on=False (or no compiled binding, no gradients): the modules run torch's
    own forward and backward."""
    convs = _convs(modules)
    ext = _nat.torch_ext()
    live = [m for m in convs if m.weight.requires_grad and m.weight.is_cuda and m.weight.dtype == torch.float32] \
        if (on and ext is not None and torch.is_grad_enabled()) else []
    for m in convs:
        m.__dict__.pop("_fv2p_gated_weight", None)
    if not live:
        return 0
    for m, w in zip(live, ext.gate_weights([m.weight for m in live])):
        if "forward" not in m.__dict__:
            m.forward = types.MethodType(_conv_forward if type(m) is nn.Conv2d else _deconv_forward, m)
        m.__dict__["_fv2p_gated_weight"] = w
    return len(live)
