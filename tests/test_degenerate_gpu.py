"""Empty and degenerate inputs across the op families (the reference either early-returns on them or kills the process: helper_launch.h:17
throws on N <= 0, the pointnet2 / iou3d wrappers exit(-1) on launch errors).  Here every op answers an empty input with an empty,
correctly shaped output or a Python exception - never a crash, a hang or an out-of-bounds write (each call is followed by a
synchronise so that a faulting kernel would surface in its own test)."""
import numpy as np
import pytest
import torch

from pcdet.ops.iou3d_nms import iou3d_nms_utils
from pcdet.ops.pointnet2.pointnet2_batch import pointnet2_utils as bu
from pcdet.ops.pointnet2.pointnet2_stack import pointnet2_utils as su
from pcdet.ops.roiaware_pool3d import roiaware_pool3d_utils
from pcdet.ops.roipoint_pool3d.roipoint_pool3d_utils import RoIPointPool3d

pytestmark = pytest.mark.gpu


def sync():
    torch.cuda.synchronize()


def test_nms_and_iou_with_no_boxes(gpu):
    boxes = torch.zeros((0, 7), device=gpu)
    keep, _ = iou3d_nms_utils.nms_gpu(boxes, torch.zeros(0, device=gpu), 0.7)
    sync()
    assert keep.numel() == 0
    some = torch.tensor([[0, 0, 0, 4, 2, 1.5, 0.3], [10, 0, 0, 4, 2, 1.5, 0.0]], device=gpu)
    a = iou3d_nms_utils.boxes_iou3d_gpu(some, boxes)
    b = iou3d_nms_utils.boxes_iou3d_gpu(boxes, some)
    c = iou3d_nms_utils.boxes_iou_bev(boxes, boxes)
    sync()
    assert tuple(a.shape) == (2, 0) and tuple(b.shape) == (0, 2) and tuple(c.shape) == (0, 0)
    one, _ = iou3d_nms_utils.nms_gpu(some[:1], torch.ones(1, device=gpu), 0.7)
    sync()
    assert one.tolist() == [0]
    # degenerate boxes: zero extent never overlaps, identical boxes suppress each other
    flat = some.clone()
    flat[:, 3:6] = 0
    assert float(iou3d_nms_utils.boxes_iou_bev(flat, flat).max()) == 0.0
    twin = some[:1].repeat(3, 1)
    k, _ = iou3d_nms_utils.nms_gpu(twin, torch.tensor([0.5, 0.9, 0.7], device=gpu), 0.5)
    sync()
    assert k.tolist() == [1]


def test_point_queries_with_no_queries_or_no_points(gpu):
    xyz = torch.rand(2, 100, 3, device=gpu)
    none = torch.zeros(2, 0, 3, device=gpu)
    idx = bu.ball_query(0.5, 8, xyz, none)
    sync()
    assert tuple(idx.shape) == (2, 0, 8)
    d, i = bu.three_nn(none, xyz)
    sync()
    assert tuple(d.shape) == (2, 0, 3) and tuple(i.shape) == (2, 0, 3)
    # stacked form with one empty sample in the middle of the batch
    unknown = torch.rand(30, 3, device=gpu)
    known = torch.rand(50, 3, device=gpu)
    d, i = su.three_nn(unknown, torch.tensor([10, 0, 20], dtype=torch.int32, device=gpu), known, torch.tensor([20, 5, 25], dtype=torch.int32, device=gpu))
    sync()
    assert bool((i[:10] < 20).all()) and bool((i[10:] >= 25).all()) and bool(torch.isfinite(d).all())
    feats = torch.randn(50, 8, device=gpu, requires_grad=True)
    w = torch.full((30, 3), 1.0 / 3.0, device=gpu)
    out = su.three_interpolate(feats, i, w)
    out.sum().backward()
    sync()
    assert tuple(out.shape) == (30, 8) and bool(torch.isfinite(feats.grad).all())
    assert float(feats.grad[20:25].abs().sum()) == 0.0          # the sample without queries reads none of its known points
    empty_out = su.three_interpolate(feats, torch.zeros((0, 3), dtype=torch.int32, device=gpu), torch.zeros((0, 3), device=gpu))
    sync()
    assert tuple(empty_out.shape) == (0, 8)
    picks = bu.furthest_point_sample(xyz, 100)
    sync()
    assert all(sorted(p.tolist()) == list(range(100)) for p in picks)


def test_roi_ops_with_no_boxes_or_no_points(gpu):
    pts = torch.rand(2, 500, 3, device=gpu) * 10
    none = torch.zeros(2, 0, 7, device=gpu)
    owner = roiaware_pool3d_utils.points_in_boxes_gpu(pts, none)
    sync()
    assert tuple(owner.shape) == (2, 500) and bool((owner == -1).all())
    box = torch.tensor([[[5, 5, 5, 2, 2, 2, 0.0]]], device=gpu).repeat(2, 1, 1)
    owner0 = roiaware_pool3d_utils.points_in_boxes_gpu(torch.zeros(2, 0, 3, device=gpu), box)
    sync()
    assert tuple(owner0.shape) == (2, 0)
    pool = RoIPointPool3d(64, 0.0)
    with torch.no_grad():
        pooled, flag = pool(pts, torch.randn(2, 500, 5, device=gpu), none)
        sync()
        assert tuple(pooled.shape) == (2, 0, 64, 8) and tuple(flag.shape) == (2, 0)
        far = box.clone()
        far[..., :3] = 1000.0
        pooled, flag = pool(pts, torch.randn(2, 500, 5, device=gpu), far)
        sync()
        assert bool((flag == 1).all()) and float(pooled.abs().sum()) == 0.0


def test_batchnorm_on_one_and_two_rows(gpu):
    """One row in training mode: the reference's nn.BatchNorm1d refuses it (ValueError) and so does the fused layer; two rows are the
    smallest batch: the fused kernels against torch's own BatchNorm1d + ReLU, forward and input gradient."""
    from fv2p_harness.backbone import bn_act
    bn = torch.nn.BatchNorm1d(16, eps=1e-3, momentum=0.01).to(gpu)
    with pytest.raises(ValueError):
        bn_act(bn, torch.randn(1, 16, device=gpu), torch.nn.ReLU())
    x = torch.randn(2, 16, device=gpu)
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    bn2 = torch.nn.BatchNorm1d(16, eps=1e-3, momentum=0.01).to(gpu)
    ya = bn_act(bn, xa, torch.nn.ReLU())
    yb = torch.relu(bn2(xb))
    g = torch.randn_like(ya)
    ya.backward(g)
    yb.backward(g)
    sync()
    assert torch.allclose(ya, yb, atol=1e-5) and torch.allclose(xa.grad, xb.grad, atol=1e-4)


def test_dcn_with_an_empty_batch(gpu):
    from pcdet.ops.DeformableConvolutionV2PyTorch.modules.modulated_deform_conv import ModulatedDeformConv
    m = ModulatedDeformConv(16, 16, 3, stride=1, padding=1, bias=True).to(gpu)
    x = torch.zeros(0, 16, 8, 8, device=gpu, requires_grad=True)
    y = m(x, torch.zeros(0, 18, 8, 8, device=gpu), torch.zeros(0, 9, 8, 8, device=gpu))
    y.sum().backward()
    sync()
    assert tuple(y.shape) == (0, 16, 8, 8) and float(m.weight.grad.abs().sum()) == 0.0
