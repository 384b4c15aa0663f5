"""GPU parity of the bilinear BEV gather (csrc/bev_interp.hip) through the mirrored Python API
(pcdet.models.backbones_3d.pfe.bev_grid_pooling) against (1) the golden outputs of the reference's own function and
(2) oracle/bev_oracle.py at the FV2P size ([B, 128, 200, 176] map, 27 648 key points per sample).
Forward: bit-exact (same fp32 operations in the same order, no contraction).  Map gradient: 1e-5 relative (float atomics
add the contributions of points that share a pixel in arbitrary order, as torch's index backward does)."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import bev_oracle
from pcdet.models.backbones_3d.pfe import bev_grid_pooling as bgp

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "bev_interp_*.npz"))),
                         ids=lambda p: os.path.basename(p)[:-4])
def test_reference_signature_on_golden_vectors(gpu, path):
    d = np.load(path)
    out = bgp.bilinear_interpolate_torch(torch.from_numpy(d["im"]).to(gpu), torch.from_numpy(d["x"]).to(gpu), torch.from_numpy(d["y"]).to(gpu))
    assert out.shape == d["out"].shape and np.array_equal(out.cpu().numpy(), d["out"])


@pytest.mark.parametrize("b,c,h,w,n", [(2, 128, 200, 176, 27648), (3, 20, 37, 29, 1000), (1, 7, 5, 4, 64)])
def test_batched_channel_first_forward_and_map_gradient(gpu, b, c, h, w, n):
    rng = np.random.default_rng(b * 100 + c)
    pc_range, voxel, stride = [0.0, -40.0, -3.0, 70.4, 40.0, 1.0], [0.05, 0.05, 0.1], 8
    bev = rng.standard_normal((b, c, h, w)).astype(np.float32)
    kp = np.stack([rng.uniform(pc_range[0] - 1.0, pc_range[0] + w * stride * voxel[0] + 1.0, (b, n)),
                   rng.uniform(pc_range[1] - 1.0, pc_range[1] + h * stride * voxel[1] + 1.0, (b, n)),
                   rng.uniform(-3, 1, (b, n))], axis=2).astype(np.float32)
    bev_t = torch.from_numpy(bev).to(gpu).requires_grad_(True)
    out = bgp.interpolate_from_bev_features(torch.from_numpy(kp).to(gpu), bev_t, b, stride, pc_range, voxel)
    # the coordinate arithmetic is torch's on the GPU (tensor / python scalar = multiply by the fp32 reciprocal): the oracle
    # restates exactly that
    kp_t = torch.from_numpy(kp).to(gpu)
    xs, ys = bev_oracle.pixel_coordinates(kp, stride, pc_range, voxel)
    assert np.array_equal((((kp_t[:, :, 0] - pc_range[0]) / voxel[0]) / stride).cpu().numpy(), xs)
    want = bev_oracle.interpolate_from_bev_features(kp, bev, stride, pc_range, voxel)
    assert out.shape == (b, n, c) and np.array_equal(out.detach().cpu().numpy(), want)
    g = rng.standard_normal((b, n, c)).astype(np.float32)
    out.backward(torch.from_numpy(g).to(gpu))
    for k in range(b):
        gk = bev_oracle.bilinear_interpolate_grad((h, w, c), xs[k], ys[k], g[k])            # (H, W, C) float64
        got = bev_t.grad[k].permute(1, 2, 0).double().cpu().numpy()
        assert np.abs(got - gk).max() / np.abs(gk).max() < 1e-5


def test_module_mirror_and_errors(gpu):
    """BEVGridPooling with the reference's constructor arguments; CPU tensors raise (no fallback)."""
    from types import SimpleNamespace
    mod = bgp.BEVGridPooling(SimpleNamespace(IN_CHANNELS=16, OUT_CHANNELS=8), [0.0, -40.0, -3.0, 70.4, 40.0, 1.0], [0.05, 0.05, 0.1]).to(gpu)
    bev = torch.randn(2, 16, 25, 22, device=gpu)
    kp = torch.rand(2, 300, 3, device=gpu) * torch.tensor([8.0, 8.0, 1.0], device=gpu) + torch.tensor([0.0, -40.0, -2.0], device=gpu)
    out = mod({"spatial_features_before_head": bev, "spatial_features_stride": 8, "batch_size": 2}, kp)
    assert out.shape == (2, 300, 8) and bool(torch.isfinite(out).all())
    import fv2p_native
    with pytest.raises(fv2p_native.Fv2pError):
        bgp.bilinear_interpolate_torch(torch.randn(4, 4, 3), torch.rand(5), torch.rand(5))


@pytest.mark.gpu
@pytest.mark.parametrize("b,r,s", [(4, 128, 35200), (2, 50, 77), (1, 3, 1), (3, 64, 130), (1, 257, 63)])
def test_transpose_batched_is_a_permutation(gpu, b, r, s):
    """fv2p_transpose_batched ([B][R][S] -> [B][S][R], the NCHW <-> NHWC copies of the DCN layers and the BEV pooling): every element
    lands where permute + contiguous puts it, for sizes that are and are not multiples of the 64 x 64 tile / of four."""
    import fv2p_native
    x = torch.randn(b, r, s, device=gpu)
    out = torch.full((b, s, r), float("nan"), device=gpu)
    fv2p_native.call("fv2p_transpose_batched", x, b, r, s, out, fv2p_native.stream())
    assert torch.equal(out, x.permute(0, 2, 1).contiguous())
