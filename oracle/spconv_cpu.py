"""ORACLE — test infrastructure only.

CPU mirror of a network built from pcdet.ops.spconv modules: every SparseConvolution is replaced by a shim
that runs the reference algorithm restated in oracle/ (dense-grid rulebook of geometry.h + per-offset
gather -> mm -> scatter-add of spconv_ops.h:260-457) on the host, sharing the layer's weights.  Used as the
end-to-end parity checker and as bench.py's cpu_baseline ("port")."""
import copy

import numpy as np
import torch
from torch.autograd import Function

import oracle
from pcdet.ops import spconv
from pcdet.ops.spconv.conv import SparseConvolution


class _ConvFn(Function):
    @staticmethod
    def forward(ctx, feats, w, pairs, num, n_out, inverse, subm):
        ctx.save_for_backward(feats, w)
        ctx.args = (pairs, num, inverse, subm)
        return oracle.indice_conv(feats.detach(), w.detach(), pairs, num, n_out, inverse=inverse, subm=subm)

    @staticmethod
    def backward(ctx, g):
        feats, w = ctx.saved_tensors
        pairs, num, inverse, subm = ctx.args
        din, dw = oracle.indice_conv_backward(feats.detach(), w.detach(), g.contiguous(), pairs, num, inverse=inverse, subm=subm)
        return din, dw, None, None, None, None, None


class CpuSparseConv(spconv.SparseModule):
    def __init__(self, src: SparseConvolution):
        super().__init__()
        self.cfg = {k: getattr(src, k) for k in ("kernel_size", "stride", "padding", "dilation", "output_padding", "subm",
                                                 "transposed", "inverse", "indice_key")}
        self.weight = torch.nn.Parameter(src.weight.detach().cpu().clone())
        self.bias = None if src.bias is None else torch.nn.Parameter(src.bias.detach().cpu().clone())

    def forward(self, x, _post=None):   # _post: the fused conv+BN offer of the product's SparseSequential, never taken here
        c = self.cfg
        ind = x.indices.numpy()
        key = c["indice_key"]
        datas = x.find_indice_pair(key)
        if c["inverse"]:
            # reference conv.py:159-166: output rows = the coupled conv's input rows
            _, in_ids, pairs, num, in_shape, _ = datas
            outids, out_shape = in_ids, in_shape
        elif key is not None and datas is not None:
            outids, _, pairs, num, _, out_shape = datas
        else:
            shape = list(x.spatial_shape)
            outids, pairs, num = oracle.indice_pairs(ind, x.batch_size, shape, c["kernel_size"], c["stride"], c["padding"],
                                                     c["dilation"], c["output_padding"], subm=c["subm"],
                                                     transpose=c["transposed"], force_sparse=True)
            if c["subm"]:
                out_shape = shape
            elif c["transposed"]:
                out_shape = oracle._deconv_out_shape(shape, c["kernel_size"], c["stride"], c["padding"], c["dilation"], c["output_padding"])
            else:
                out_shape = oracle._conv_out_shape(shape, c["kernel_size"], c["stride"], c["padding"], c["dilation"])
            x.indice_dict[key] = (outids, ind, pairs, num, shape, out_shape)
        out = _ConvFn.apply(x.features, self.weight, pairs, num, outids.shape[0], bool(c["inverse"]), bool(c["subm"]))
        if self.bias is not None:
            out = out + self.bias
        y = spconv.SparseConvTensor(out, torch.from_numpy(np.ascontiguousarray(outids)), out_shape, x.batch_size)
        y.indice_dict = x.indice_dict
        return y


def cpu_mirror(model):
    """Deep copy of `model` on the CPU with every SparseConvolution swapped for the oracle shim."""
    convs = {name: m for name, m in model.named_modules() if isinstance(m, SparseConvolution)}
    cpu = copy.deepcopy(model).cpu()
    for name, src in convs.items():
        parent = cpu
        parts = name.split(".")
        for p in parts[:-1]:
            parent = parent._modules[p]
        parent._modules[parts[-1]] = CpuSparseConv(src)
    return cpu
