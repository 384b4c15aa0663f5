"""pcdet.ops.spconv — drop-in for the reference's vendored spconv 1.x package
(pcdet/ops/spconv/__init__.py:15-40): same public names, MI355X-native implementation."""
from .conv import (SparseConv2d, SparseConv3d, SparseConv4d, SparseConvolution, SparseConvTranspose2d,
                   SparseConvTranspose3d, SparseInverseConv2d, SparseInverseConv3d, SubMConv2d, SubMConv3d, SubMConv4d, defer_weight_gradients)
from .group import SparseGroup3d, SubMGroup3d
from .modules import RemoveGrid, SparseModule, SparseSequential, ToDense
from .pool import SparseMaxPool2d, SparseMaxPool3d
from .prefetch import attach_rulebooks, build_rulebooks, rulebook_recipe
from .structure import SparseConvTensor, scatter_nd

__all__ = [
    'SparseConv2d', 'SparseConv3d', 'SubMConv2d', 'SubMConv3d', 'SparseConvTranspose2d', 'SparseConvTranspose3d',
    'SparseInverseConv2d', 'SparseInverseConv3d', 'SparseModule', 'SparseSequential', 'SparseMaxPool2d',
    'SparseMaxPool3d', 'SparseConvTensor', 'scatter_nd', 'SparseGroup3d', 'SubMGroup3d',
    'rulebook_recipe', 'build_rulebooks', 'attach_rulebooks', 'defer_weight_gradients',
]
