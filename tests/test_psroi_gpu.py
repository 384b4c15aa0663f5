"""GPU parity of deformable PSROI pooling through the reference's call surface
(DCN.deform_psroi_pooling_forward/backward, DeformRoIPoolingFunction, DeformRoIPooling, DeformRoIPoolingPack) against
oracle/psroi_oracle.py, plus the reference's two self-checks (DeformableConvolutionV2PyTorch/test.py:437-468, :471-505).

Tolerances: counts bit-exact; pooled values 1e-5 (same fp32 operations in the same order on both sides); gradients 1e-4
relative to the largest entry (atomic accumulation order differs from the oracle's float64 sums)."""
import numpy as np
import pytest
import torch

import fv2p_native
from oracle import psroi_oracle as ps
from pcdet.ops.DeformableConvolutionV2PyTorch import DCN
from pcdet.ops.DeformableConvolutionV2PyTorch.functions import DeformRoIPoolingFunction
from pcdet.ops.DeformableConvolutionV2PyTorch.modules import DeformRoIPooling, DeformRoIPoolingPack, _DeformRoIPooling

pytestmark = pytest.mark.gpu


def case(seed, batch, channels, h, w, rois, pooled, part, classes, extent):
    rng = np.random.default_rng(seed)
    data = rng.standard_normal((batch, channels, h, w)).astype(np.float32)
    bi = rng.integers(0, batch, (rois, 1))
    x, y = rng.random((rois, 1)) * extent[0] - 8, rng.random((rois, 1)) * extent[1] - 8      # some RoIs start off the map
    bw, bh = rng.random((rois, 1)) * extent[0] * 0.4, rng.random((rois, 1)) * extent[1] * 0.4
    boxes = np.concatenate([bi, x, y, x + bw, y + bh], 1).astype(np.float32)
    trans = rng.standard_normal((rois, 2 * classes, part, part)).astype(np.float32)
    return data, boxes, trans


CASES = [  # batch, channels, H, W, rois, pooled, part, classes, spp, trans_std, scale
    (2, 3, 5, 5, 4, 3, 3, 1, 4, 0.1, 0.25),            # the reference's gradcheck shapes
    (2, 16, 64, 64, 12, 7, 7, 1, 4, 0.1, 0.25),        # the reference's example shapes (test.py:533-574)
    (3, 64, 50, 44, 128, 7, 7, 4, 2, 0.2, 0.125),      # a BEV-map sized case, 4 shift classes
    (1, 8, 33, 17, 9, 5, 3, 2, 3, 0.3, 0.5),           # part_size != pooled_size, odd map
]


def test_samples_on_bin_borders_and_on_the_map_limits(gpu):
    """RoIs whose samples fall EXACTLY on the limits of the inside test (w == -0.5, w == width - 0.5) and on integer pixel positions
    (floor == ceil corner): integer corners on a power-of-two lattice, so every position is exact in float32 and in float64.  The
    reference evaluates these tests with double literals (deform_psroi_pooling_cuda.cu:131-136), the kernel in float32 (psroi.hip
    header: each is one correctly rounded operation) — counts and values must agree with the oracle run in either precision."""
    h = w = 16
    data = np.random.default_rng(5).standard_normal((1, 2, h, w)).astype(np.float32)
    boxes = np.array([[0, 0, 0, 15, 15],        # start -0.5, end 7.5: bins of 2, samples -0.5, 0.5, ... (first one ON the lower limit)
                      [0, 1, 1, 31, 31],        # start 0, end 15.5 (ON the upper limit of a 16-wide map after scaling)
                      [0, -2, -2, 13, 13],      # start -1.5: first samples outside, then -0.5 exactly
                      [0, 18, 18, 33, 33],      # start 8.5, unit steps: the last sample ON the upper limit 15.5 (inside)
                      [0, 20, 20, 35, 35],      # start 9.5: the last sample at 16.5 (outside)
                      [0, 7, 7, 7, 7]], np.float32)   # one pixel: 0.5 wide, bins of 0.125
    conf = (1, 0.5, 2, 1, 4, 4, 2, 0.0)
    want32, cnt32 = ps.deform_psroi_pooling_forward(data, boxes, None, True, *conf[1:])
    want64, cnt64 = ps.deform_psroi_pooling_forward(data.astype(np.float64), boxes.astype(np.float64), None, True, *conf[1:])
    assert np.array_equal(cnt32, cnt64)
    d, b = torch.from_numpy(data).to(gpu), torch.from_numpy(boxes).to(gpu)
    out, cnt = DCN.deform_psroi_pooling_forward(d, b, d.new(), *conf)
    assert np.array_equal(cnt.cpu().numpy(), cnt64), "inside test on the limits differs"
    assert 0 < (cnt64 < 4).mean() < 1
    assert np.abs(out.cpu().numpy() - want64).max() < 1e-6


@pytest.mark.parametrize("cfg", CASES)
@pytest.mark.parametrize("no_trans", [False, True])
def test_forward_and_backward_match_the_oracle(gpu, cfg, no_trans):
    batch, channels, h, w, rois, pooled, part, classes, spp, tstd, scale = cfg
    data, boxes, trans = case(sum(cfg[:5]), batch, channels, h, w, rois, pooled, part, classes, (w / scale, h / scale))
    conf = (int(no_trans), scale, channels, 1, pooled, part, spp, tstd)
    want, want_cnt = ps.deform_psroi_pooling_forward(data, boxes, trans, no_trans, *conf[1:])
    d, b, t = (torch.from_numpy(a).to(gpu) for a in (data, boxes, trans))
    if no_trans:
        t = d.new()
    out, cnt = DCN.deform_psroi_pooling_forward(d, b, t, *conf)
    assert out.shape == want.shape and out.dtype == torch.float32
    assert np.array_equal(cnt.cpu().numpy(), want_cnt), "sample counts differ"
    assert 0 < (want_cnt < spp * spp).mean() < 1, "the case must hold clipped and unclipped bins"
    assert np.abs(out.cpu().numpy() - want).max() < 1e-5
    g = np.random.default_rng(7).standard_normal(want.shape).astype(np.float32)
    gd, gt = DCN.deform_psroi_pooling_backward(torch.from_numpy(g).to(gpu), d, b, t, cnt, *conf)
    wd, wt = ps.deform_psroi_pooling_backward(g, data, boxes, None if no_trans else trans, want_cnt, no_trans, *conf[1:])
    assert np.abs(gd.cpu().numpy() - wd).max() < 1e-4 * np.abs(wd).max()
    if no_trans:
        assert gt.numel() == 0
    else:
        assert gt.shape == t.shape and np.abs(gt.cpu().numpy() - wt).max() < 1e-4 * np.abs(wt).max()


def test_pooling_zero_offset(gpu):
    """check_pooling_zero_offset (test.py:437-468): the block image; zero shifts pool exactly like no_trans; the known answer of
    tests/test_psroi_oracle.py through the oracle."""
    x = torch.zeros(2, 16, 64, 64, device=gpu)
    x[0, :, 16:26, 16:26] = 1.
    x[1, :, 10:20, 20:30] = 2.
    rois = torch.tensor([[0, 65, 65, 103, 103], [1, 81, 41, 119, 79]], device=gpu).float()
    out = DeformRoIPooling(spatial_scale=1.0 / 4, pooled_size=7, output_dim=16, no_trans=True, group_size=1, trans_std=0.0).to(gpu)(x, rois, x.new())
    dout = DeformRoIPooling(spatial_scale=1.0 / 4, pooled_size=7, output_dim=16, no_trans=False, group_size=1, trans_std=0.0).to(gpu)(
        x, rois, torch.zeros(20, 2, 7, 7, device=gpu))
    assert torch.equal(out, dout)
    want, _ = ps.deform_psroi_pooling_forward(x.cpu().numpy(), rois.cpu().numpy(), None, True, 0.25, 16, 1, 7, 7, 4, 0.0)
    assert np.array_equal(out.cpu().numpy(), want)
    assert 0.9 < float(out[0].mean()) < 1.0 and 1.8 < float(out[1].mean()) < 2.0


def test_autograd_function_gradients(gpu):
    """check_gradient_dpooling (test.py:471-505): the Function's analytic gradients against central differences of its own forward
    pass (fp32, so eps 1e-2 on a smooth random case and a loose bound) and, tightly, against the float64 oracle."""
    data, boxes, trans = case(11, 2, 3, 9, 9, 4, 3, 3, 1, (36, 36))
    boxes[:, 1:3] = np.abs(boxes[:, 1:3]) + 2      # on the map
    d = torch.from_numpy(data).to(gpu).requires_grad_(True)
    t = torch.from_numpy(trans).to(gpu).requires_grad_(True)
    b = torch.from_numpy(boxes).to(gpu)
    out = _DeformRoIPooling(d, b, t, 0.25, 3, 3, 0, 1, 3, 4, 0.1)
    g = torch.randn_like(out)
    out.backward(g)
    d64, t64 = data.astype(np.float64), trans.astype(np.float64)
    o64, c64 = ps.deform_psroi_pooling_forward(d64, boxes.astype(np.float64), t64, False, 0.25, 3, 1, 3, 3, 4, 0.1)
    wd, wt = ps.deform_psroi_pooling_backward(g.cpu().numpy().astype(np.float64), d64, boxes.astype(np.float64), t64, c64, False, 0.25, 3, 1, 3, 3, 4, 0.1)
    assert np.abs(d.grad.cpu().numpy() - wd).max() < 1e-4 * np.abs(wd).max()
    assert np.abs(t.grad.cpu().numpy() - wt).max() < 1e-4 * np.abs(wt).max()
    assert isinstance(out.grad_fn, DeformRoIPoolingFunction._backward_cls)


def test_pack_module(gpu):
    """DeformRoIPoolingPack (modules/deform_psroi_pooling.py:53-130): zero-initialised last Linear -> zero shifts and mask 0.5, so
    the module starts as half the plain pooling; its parameters receive gradients."""
    torch.manual_seed(0)
    x = torch.randn(2, 8, 32, 32, device=gpu, requires_grad=True)
    rois = torch.tensor([[0, 8, 8, 60, 70], [1, 20, 30, 100, 90], [1, 0, 0, 40, 40]], device=gpu).float()
    pack = DeformRoIPoolingPack(spatial_scale=0.25, pooled_size=5, output_dim=8, no_trans=False, group_size=1, trans_std=0.1, deform_fc_dim=32).to(gpu)
    assert list(pack.state_dict()) == ["offset_mask_fc.%d.%s" % (i, n) for i in (0, 2, 4) for n in ("weight", "bias")]
    plain = DeformRoIPooling(0.25, 5, 8, True)(x, rois, x.new())
    y = pack(x, rois)
    assert y.shape == (3, 8, 5, 5) and torch.allclose(y, 0.5 * plain, atol=1e-6)
    torch.nn.init.normal_(pack.offset_mask_fc[4].weight, std=0.05)
    y = pack(x, rois)
    y.square().sum().backward()
    assert x.grad is not None and torch.isfinite(x.grad).all()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() and p.grad.abs().sum() > 0 for p in pack.parameters())
    only = DeformRoIPoolingPack(0.25, 5, 8, True)
    assert len(list(only.parameters())) == 0 and torch.equal(only(x, rois), plain)


def test_edges_and_errors(gpu):
    x = torch.randn(1, 4, 8, 8, device=gpu)
    out, cnt = DCN.deform_psroi_pooling_forward(x, x.new_zeros((0, 5)), x.new(), 1, 0.25, 4, 1, 3, 3, 4, 0.0)
    assert out.shape == (0, 4, 3, 3) and cnt.shape == (0, 4, 3, 3)
    gd, gt = DCN.deform_psroi_pooling_backward(out, x, x.new_zeros((0, 5)), x.new(), cnt, 1, 0.25, 4, 1, 3, 3, 4, 0.0)
    assert gd.shape == x.shape and float(gd.abs().sum()) == 0
    rois = torch.tensor([[3, 0, 0, 16, 16], [0, 0, 0, 16, 16]], device=gpu).float()       # batch index 3 of a batch of 1
    out, cnt = DCN.deform_psroi_pooling_forward(x, rois, x.new(), 1, 0.25, 4, 1, 3, 3, 4, 0.0)
    assert float(out[0].abs().sum()) == 0 and float(cnt[0].sum()) == 0 and float(cnt[1].min()) > 0
    with pytest.raises(AssertionError, match="input channels and output channels must equal"):     # deform_psroi_pooling_cuda.cu:291
        DCN.deform_psroi_pooling_forward(x, rois, x.new(), 1, 0.25, 2, 1, 3, 3, 4, 0.0)
    with pytest.raises(fv2p_native.Fv2pError, match="group_size"):          # the reference reads beyond the map here
        DCN.deform_psroi_pooling_forward(x, rois, x.new(), 1, 0.25, 4, 2, 3, 3, 4, 0.0)
    with pytest.raises(fv2p_native.Fv2pError, match="CPU"):
        DCN.deform_psroi_pooling_forward(x.cpu(), rois.cpu(), x.new().cpu(), 1, 0.25, 4, 1, 3, 3, 4, 0.0)


def test_group_size_through_the_c_abi(gpu):
    """The C entry point takes the position-sensitive layout the Python surface cannot express (C = output_dim * group_size^2)."""
    rng = np.random.default_rng(5)
    g, out_dim, pooled = 2, 3, 4
    data = rng.standard_normal((2, out_dim * g * g, 12, 12)).astype(np.float32)
    rois = np.array([[0, 4, 4, 40, 40], [1, -3, 6, 30, 50]], np.float32)
    trans = rng.standard_normal((2, 2, pooled, pooled)).astype(np.float32)
    want, want_cnt = ps.deform_psroi_pooling_forward(data, rois, trans, False, 0.25, out_dim, g, pooled, pooled, 2, 0.1)
    d, r, t = (torch.from_numpy(a).to(gpu) for a in (data, rois, trans))
    out = torch.empty(want.shape, device=gpu)
    cnt = torch.empty(want.shape, device=gpu)
    fv2p_native.call("fv2p_deform_psroi_pool_forward", d, r, t, 2, out_dim * g * g, 12, 12, 2, 0, 0.25, out_dim, g, pooled, pooled, 2, 0.1, 1,
                     out, cnt, fv2p_native.stream())
    assert np.array_equal(cnt.cpu().numpy(), want_cnt) and np.abs(out.cpu().numpy() - want).max() < 1e-5
