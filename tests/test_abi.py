"""CPU: the C-ABI shared library loads and exports every symbol include/fv2p_ops.h declares."""
import ctypes
import os

import fv2p_native as nat


def test_library_is_in_tree_and_loads():
    assert os.path.exists(nat.LIB_PATH), "run __graft_entry__.build() first"
    lib = nat.lib()
    assert lib.fv2p_abi_version() == 1


def test_every_declared_symbol_is_exported():
    raw = ctypes.CDLL(nat.LIB_PATH)
    names = list(nat.declared_symbols())
    assert len(names) >= 8
    missing = [n for n in names if not hasattr(raw, n)]
    assert not missing, missing


def test_ws_queries_are_pure_host_functions():
    lib = nat.lib()
    assert lib.fv2p_points_to_voxel_ws_bytes(16384, 16000) > 0
    assert lib.fv2p_scan_ws_bytes(10) > 0
    assert lib.fv2p_radix_sort_ws_bytes(1 << 20) > 0


def test_product_package_never_imports_the_oracle():
    root = os.path.dirname(nat.__file__)
    for d, _, files in os.walk(root):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                src = open(os.path.join(d, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "liboracle" not in src, f


def test_compiled_torch_binding_loads_and_matches_the_abi():
    """lib/fv2p_torch.so (csrc_torch/fv2p_torch.cpp) imports without a GPU, links the same libfv2p_ops.so and exposes the
    two autograd ops the Python API dispatches to."""
    import fv2p_native
    ext = fv2p_native.torch_ext()
    assert ext is not None, "lib/fv2p_torch.so is missing: build with __graft_entry__.build()"
    assert ext.abi_version() == fv2p_native.lib().fv2p_abi_version() == 1
    assert callable(ext.sparse_conv) and callable(ext.batch_norm_relu)
