"""SparseGroup3d / SubMGroup3d — the fork author's "gather the K neighbours, no GEMM" modules
(reference spconv/group.py:27-196): features (N, C) -> (N_out, K, C)."""
import numpy as np

from . import functional as Fsp
from . import ops
from .modules import SparseModule
from .structure import SparseConvTensor


class SparseGroup(SparseModule):

    def __init__(self, ndim, in_channels, kernel_size=3, stride=1, padding=0, dilation=1, subm=False, output_padding=0,
                 indice_key=None):
        super(SparseGroup, self).__init__()
        as_list = lambda v: list(v) if isinstance(v, (list, tuple)) else [v] * ndim
        kernel_size, stride, padding = as_list(kernel_size), as_list(stride), as_list(padding)
        dilation, output_padding = as_list(dilation), as_list(output_padding)
        for d, s in zip(dilation, stride):
            assert any([s == 1, d == 1]), "don't support this."
        self.ndim = ndim
        self.in_channels = in_channels
        self.kernel_size = kernel_size
        self.group1x1 = np.prod(kernel_size) == 1
        self.stride = stride
        self.padding = padding
        self.dilation = dilation
        self.output_padding = output_padding
        self.subm = subm
        self.indice_key = indice_key

    def forward(self, input):
        assert isinstance(input, SparseConvTensor)
        features = input.features
        assert len(features.shape) == 2 and features.shape[1] == self.in_channels
        if self.group1x1:
            # (N, C) -> (N, 1, C).  (The reference spells this `unsequeeze` and would raise, group.py:95.)
            out_tensor = SparseConvTensor(features.unsqueeze(dim=1), input.indices, input.spatial_shape, input.batch_size)
            out_tensor.indice_dict = input.indice_dict
            out_tensor.grid = input.grid
            return out_tensor
        rb = input.find_indice_pair(self.indice_key)
        if self.indice_key is None or rb is None:
            rb = ops.build_rulebook(input.indices, input.batch_size, input.spatial_shape, self.kernel_size, self.stride,
                                    self.padding, self.dilation, self.output_padding, self.subm, False)
            input.indice_dict[self.indice_key] = rb
        fn = Fsp.indice_subm_group if self.subm else Fsp.indice_group
        out_features = fn(features, rb, None, rb.outids.shape[0]).permute(1, 0, 2)
        out_shape = input.spatial_shape if self.subm else rb.out_spatial_shape
        out_tensor = SparseConvTensor(out_features, rb.outids, out_shape, input.batch_size)
        out_tensor.indice_dict = input.indice_dict
        out_tensor.grid = input.grid
        return out_tensor


class SparseGroup3d(SparseGroup):

    def __init__(self, in_channels, kernel_size, stride=1, padding=0, dilation=1, indice_key=None):
        super(SparseGroup3d, self).__init__(3, in_channels, kernel_size, stride, padding, dilation, indice_key=indice_key)


class SubMGroup3d(SparseGroup):

    def __init__(self, in_channels, kernel_size, stride=1, padding=0, dilation=1, indice_key=None):
        super(SubMGroup3d, self).__init__(3, in_channels, kernel_size, stride, padding, dilation, True,
                                          indice_key=indice_key)
