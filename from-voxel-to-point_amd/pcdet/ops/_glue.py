"""Shared plumbing of the Python ops layer: one helper that issues a C-ABI call of libfv2p_ops.so on the caller's stream and
one that turns a pair of plain functions into a torch.autograd.Function.  The op modules next to this file describe each
operator as `forward(saved, *inputs)` / `backward(saved, *grads)` on top of these two; nothing here falls back to the CPU."""
import torch
from torch.autograd import Function

import fv2p_native as _nat


def run(symbol, *args):
    """Validates the tensor arguments (device, contiguity) and calls `symbol` with the current stream appended."""
    tensors = [a for a in args if torch.is_tensor(a)]
    _nat.require_cuda(*tensors)
    for t in tensors:
        if not t.is_contiguous():
            raise _nat.Fv2pError(f"{symbol}: tensors must be contiguous")
    with _nat.device_guard(tensors[0].device):
        return _nat.call(symbol, *args, _nat.stream())


def scratch(nbytes_symbol, device, *size_args):
    """Workspace tensor of the size the library asks for (grow-only, per device and stream)."""
    with _nat.device_guard(device):
        return _nat.workspace(getattr(_nat.lib(), nbytes_symbol)(*size_args), device)


def new(like, shape, dtype=torch.float32, fill=None):
    t = torch.empty(shape, dtype=dtype, device=like.device)
    return t if fill is None else t.fill_(fill)


def autograd_op(name, forward, backward=None, doc=None):
    """class `name`(Function) with forward(ctx, *inputs) = forward(ctx.saved, *inputs) and, when `backward` is given,
    backward(ctx, *grads) = backward(ctx.saved, *grads); without it every input gets a None gradient (index-valued ops)."""

    def _forward(ctx, *inputs):
        ctx.saved = {}
        ctx.n_inputs = len(inputs)
        return forward(ctx.saved, *inputs)

    def _backward(ctx, *grads):
        if backward is None:
            return (None,) * ctx.n_inputs
        out = backward(ctx.saved, *grads)
        out = out if isinstance(out, tuple) else (out,)
        return out + (None,) * (ctx.n_inputs - len(out))

    return type(name, (Function,), {"forward": staticmethod(_forward), "backward": staticmethod(_backward), "__doc__": doc or forward.__doc__})
