"""SparseConvTensor container — same surface as the reference's spconv/structure.py:5-71."""
import math

import numpy as np
import torch


def scatter_nd(indices, updates, shape):
    """Dense tensor of `shape` with `updates` written at `indices` (no duplicate handling),
    cf. reference structure.py:5-18."""
    ret = torch.zeros(*shape, dtype=updates.dtype, device=updates.device)
    ndim = indices.shape[-1]
    flat = indices.reshape(-1, ndim)
    out_shape = list(indices.shape[:-1]) + list(shape[ndim:])
    ret[tuple(flat[:, i] for i in range(ndim)) + (Ellipsis,)] = updates.reshape(*out_shape)
    return ret


class _Dense(torch.autograd.Function):
    """features [N, C] -> dense [B, C, *spatial] (or [B, *spatial, C]); gradient = gather at the active cells."""

    @staticmethod
    def forward(ctx, features, indices, spatial, batch, channels_first):
        import fv2p_native as _nat
        n, c = features.shape
        ndim = len(spatial)
        shape = [batch, c] + list(spatial) if channels_first else [batch] + list(spatial) + [c]
        out = torch.empty(shape, dtype=torch.float32, device=features.device)
        sp3 = list(spatial) + [1] * (3 - ndim)
        ind = indices.contiguous()
        with _nat.device_guard(features.device):
            _nat.call("fv2p_sparse_to_dense", features.contiguous(), ind, n, c, ndim, batch, sp3, int(channels_first), out, _nat.stream())
        ctx.save_for_backward(ind)
        ctx.meta = (n, c, ndim, batch, sp3, channels_first)
        return out

    @staticmethod
    def backward(ctx, grad):
        import fv2p_native as _nat
        (ind,) = ctx.saved_tensors
        n, c, ndim, batch, sp3, channels_first = ctx.meta
        rows = torch.empty((n, c), dtype=torch.float32, device=grad.device)
        with _nat.device_guard(grad.device):
            _nat.call("fv2p_dense_to_sparse", grad.contiguous(), ind, n, c, ndim, batch, sp3, int(channels_first), rows, _nat.stream())
        return rows, None, None, None, None


class SparseConvTensor(object):
    """features [N,C] f32, indices [N,1+ndim] i32 (batch, z, y, x), spatial_shape, batch_size.

    `indice_dict` caches rulebooks per indice_key and is shared by reference between the tensors
    of one network pass (reference conv.py:227); `grid` is kept for API compatibility only — the
    hashed rulebook builder never needs a pre-allocated dense grid."""

    def __init__(self, features, indices, spatial_shape, batch_size, grid=None):
        self.features = features
        self.indices = indices if indices.dtype == torch.int32 else indices.int()
        self.spatial_shape = spatial_shape
        self.batch_size = batch_size
        # rulebooks prefetched for exactly this coordinate tensor (prefetch.attach_rulebooks), else empty
        pre = getattr(indices, "_fv2p_indice_dict", None)
        self.indice_dict = dict(pre) if pre else {}
        self.grid = grid

    @property
    def spatial_size(self):
        return np.prod(self.spatial_shape)

    def find_indice_pair(self, key):
        if key is None:
            return None
        return self.indice_dict.get(key, None)

    def dense(self, channels_first=True):
        ndim = len(self.spatial_shape)
        f = self.features
        if f.is_cuda and f.dtype == torch.float32 and f.dim() == 2 and ndim in (2, 3) and not torch.is_autocast_enabled():
            # one fill + one scatter straight into the requested layout (csrc/sparse_aux.hip), instead of
            # zeros -> index scatter -> permute -> contiguous over the whole dense volume
            return _Dense.apply(f, self.indices, tuple(int(v) for v in self.spatial_shape), int(self.batch_size), bool(channels_first))
        out_shape = [self.batch_size] + list(self.spatial_shape) + [self.features.shape[1]]
        res = scatter_nd(self.indices.long(), self.features, out_shape)
        if not channels_first:
            return res
        ndim = len(self.spatial_shape)
        perm = list(range(0, ndim + 1))
        perm.insert(1, ndim + 1)
        return res.permute(*perm).contiguous()

    @property
    def sparity(self):
        return self.indices.shape[0] / math.prod(int(v) for v in self.spatial_shape) / self.batch_size   # python ints: called per layer
