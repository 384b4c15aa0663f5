"""SparseMaxPool{2d,3d} — surface of reference spconv/pool.py:21-85 (output = max(0, neighbours))."""
from . import functional as Fsp
from . import ops
from .modules import SparseModule
from .structure import SparseConvTensor


class SparseMaxPool(SparseModule):

    def __init__(self, ndim, kernel_size, stride=1, padding=0, dilation=1, subm=False):
        super(SparseMaxPool, self).__init__()
        as_list = lambda v: list(v) if isinstance(v, (list, tuple)) else [v] * ndim
        self.ndim = ndim
        self.kernel_size = as_list(kernel_size)
        self.stride = as_list(stride)
        self.padding = as_list(padding)
        self.subm = subm
        self.dilation = as_list(dilation)

    def forward(self, input):
        assert isinstance(input, SparseConvTensor)
        rb = ops.build_rulebook(input.indices, input.batch_size, input.spatial_shape, self.kernel_size, self.stride,
                                self.padding, self.dilation, 0, self.subm)
        out_features = Fsp.indice_maxpool(input.features, rb, None, rb.outids.shape[0])
        out_tensor = SparseConvTensor(out_features, rb.outids, rb.out_spatial_shape, input.batch_size)
        out_tensor.indice_dict = input.indice_dict
        out_tensor.grid = input.grid
        return out_tensor


class SparseMaxPool2d(SparseMaxPool):

    def __init__(self, kernel_size, stride=1, padding=0, dilation=1):
        super(SparseMaxPool2d, self).__init__(2, kernel_size, stride, padding, dilation)


class SparseMaxPool3d(SparseMaxPool):

    def __init__(self, kernel_size, stride=1, padding=0, dilation=1):
        super(SparseMaxPool3d, self).__init__(3, kernel_size, stride, padding, dilation)
