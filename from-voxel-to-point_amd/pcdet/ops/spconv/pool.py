"""SparseMaxPool2d / SparseMaxPool3d — surface of the reference's spconv/pool.py:21-85.  A pooled row is the maximum over the
active inputs its kernel window covers, starting from zero (the reference zero-fills the output, pool_ops.h:25-57), so features
below zero pool to zero."""
from . import functional as Fsp
from . import ops
from .modules import SparseModule
from .structure import SparseConvTensor


class SparseMaxPool(SparseModule):
    """forward(SparseConvTensor) -> SparseConvTensor on the rulebook of a regular (or, with subm, submanifold) convolution window."""

    def __init__(self, ndim, kernel_size, stride=1, padding=0, dilation=1, subm=False):
        super().__init__()
        per_dim = lambda v: list(v) if isinstance(v, (list, tuple)) else [v] * ndim
        self.ndim, self.subm = ndim, subm
        self.kernel_size, self.stride, self.padding, self.dilation = (per_dim(v) for v in (kernel_size, stride, padding, dilation))

    def forward(self, input):
        assert isinstance(input, SparseConvTensor)
        book = ops.build_rulebook(input.indices, input.batch_size, input.spatial_shape, self.kernel_size, self.stride, self.padding,
                                  self.dilation, 0, self.subm)
        pooled = Fsp.indice_maxpool(input.features, book, None, book.outids.shape[0])
        out = SparseConvTensor(pooled, book.outids, book.out_spatial_shape, input.batch_size)
        out.indice_dict, out.grid = input.indice_dict, input.grid
        return out


def _fixed_ndim(name, ndim):
    def __init__(self, kernel_size, stride=1, padding=0, dilation=1):
        SparseMaxPool.__init__(self, ndim, kernel_size, stride, padding, dilation)
    return type(name, (SparseMaxPool,), {"__init__": __init__, "__module__": __name__})


SparseMaxPool2d = _fixed_ndim("SparseMaxPool2d", 2)
SparseMaxPool3d = _fixed_ndim("SparseMaxPool3d", 3)
