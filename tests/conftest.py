import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "from-voxel-to-point_amd")
for p in (REPO, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # GPU tests never silently pass on a box without a GPU: they are skipped unless selected with -m gpu,
    # and when selected they fail loudly if CUDA/HIP is unavailable (see `gpu` fixture).
    pass


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def gpu():
    import torch
    assert torch.cuda.is_available(), "GPU test selected but no HIP device is visible"
    import fv2p_native
    fv2p_native.lib()  # raises if libfv2p_ops.so is missing: no fallback
    return torch.device("cuda:0")


@pytest.fixture(params=["compiled", "ctypes"])
def front_end(request):
    """Runs a test once per Python front end of the library: the compiled autograd binding (lib/fv2p_torch.so) and the
    ctypes binding with Python autograd Functions.  Both reach the same kernels through the C ABI."""
    import fv2p_native
    ext = fv2p_native.torch_ext()
    assert ext is not None, "lib/fv2p_torch.so is missing: build with __graft_entry__.build()"
    if request.param == "ctypes":
        fv2p_native._EXT = None
    try:
        yield request.param
    finally:
        fv2p_native._EXT = ext
