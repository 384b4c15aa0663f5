"""pcdet.ops.spconv — drop-in for the reference's vendored spconv 1.x package (pcdet/ops/spconv/__init__.py:15-40): the same public
names, on the HIP kernels of libfv2p_ops.  Additions of this package sit at the end of `__all__`."""
from . import conv as _conv
from . import group as _group
from . import modules as _modules
from . import pool as _pool
from . import prefetch as _prefetch
from . import structure as _structure

_REFERENCE_NAMES = {
    _conv: ("SparseConv2d", "SparseConv3d", "SparseConv4d", "SubMConv2d", "SubMConv3d", "SubMConv4d", "SparseConvTranspose2d",
            "SparseConvTranspose3d", "SparseInverseConv2d", "SparseInverseConv3d", "SparseConvolution"),
    _modules: ("SparseModule", "SparseSequential", "ToDense", "RemoveGrid"),
    _pool: ("SparseMaxPool2d", "SparseMaxPool3d"),
    _structure: ("SparseConvTensor", "scatter_nd"),
    _group: ("SparseGroup3d", "SubMGroup3d"),
}
_OWN_NAMES = {_prefetch: ("rulebook_recipe", "build_rulebooks", "attach_rulebooks"), _conv: ("defer_weight_gradients", "conv_bn_fold", "materialise_pending", "set_bn_fold", "fold_enabled")}

__all__ = []
for _table in (_REFERENCE_NAMES, _OWN_NAMES):
    for _mod, _names in _table.items():
        for _name in _names:
            globals()[_name] = getattr(_mod, _name)
            __all__.append(_name)
del _table, _mod, _names, _name
