"""GPU parity of points_in_boxes / RoIPointPool3d / RoIAwarePool3d (A15, A18) against the oracle.

Bar: all index outputs bit-exact (box index per point, pooled point order, voxel point lists, argmax);
pooled float features are copies / max -> bit-exact; avg pooling within 1e-6."""
import numpy as np
import pytest
import torch

import oracle
from boxes_util import random_boxes
from fv2p_harness import synth
from pcdet.ops.roiaware_pool3d import roiaware_pool3d_utils
from pcdet.ops.roipoint_pool3d.roipoint_pool3d_utils import RoIPointPool3d

pytestmark = pytest.mark.gpu


def scene(seed, n_pts, batch):
    pts, boxes = [], []
    for b in range(batch):
        p, bx = synth.lidar_cloud(seed + b, n_pts, return_boxes=True)
        pts.append(p[:, :3])
        boxes.append(bx[:20])
    return np.stack(pts).astype(np.float32), np.stack(boxes).astype(np.float32)


def test_points_in_boxes_gpu_and_cpu(gpu):
    pts, boxes = scene(0, 16384, 2)
    out = roiaware_pool3d_utils.points_in_boxes_gpu(torch.from_numpy(pts).to(gpu), torch.from_numpy(boxes).to(gpu))
    ref = oracle.points_in_boxes_gpu(pts, boxes)
    assert out.dtype == torch.int32 and np.array_equal(out.cpu().numpy(), ref)
    assert (ref >= 0).sum() > 100  # the synthetic cars do contain LiDAR returns
    # axis-aligned closed form: a point is inside iff |d| <= half extents (away from the 1e-5 margin)
    box = np.array([[[10, 0, 0, 4, 2, 1.5, 0.0]]], np.float32)
    rng = np.random.default_rng(0)
    p = (rng.uniform(-4, 4, size=(1, 5000, 3)) + np.array([10, 0, 0])).astype(np.float32)
    got = roiaware_pool3d_utils.points_in_boxes_gpu(torch.from_numpy(p).to(gpu), torch.from_numpy(box).to(gpu)).cpu().numpy()[0]
    d = np.abs(p[0] - np.array([10, 0, 0], np.float32))
    clear = (np.abs(d[:, 0] - 2) > 1e-3) & (np.abs(d[:, 1] - 1) > 1e-3) & (np.abs(d[:, 2] - 0.75) > 1e-3)
    inside = (d[:, 0] < 2) & (d[:, 1] < 1) & (d[:, 2] < 0.75)
    assert np.array_equal((got == 0)[clear], inside[clear])
    # overlapping boxes: the first box wins (kernel breaks at the first hit)
    two = np.array([[[10, 0, 0, 4, 2, 1.5, 0.3], [10, 0, 0, 4, 2, 1.5, 0.3]]], np.float32)
    got2 = roiaware_pool3d_utils.points_in_boxes_gpu(torch.from_numpy(p).to(gpu), torch.from_numpy(two).to(gpu)).cpu().numpy()
    assert set(np.unique(got2)) <= {-1, 0}
    # CPU entry point (MARGIN 1e-2), numpy in / numpy out
    m = roiaware_pool3d_utils.points_in_boxes_cpu(pts[0][:3000], boxes[0])
    assert m.shape == (20, 3000) and np.array_equal(m, oracle.points_in_boxes_cpu(pts[0][:3000], boxes[0]))


@pytest.mark.parametrize("width", [[1.0, 1.0, 1.0], 0.4])
def test_roipoint_pool3d(gpu, width):
    """FV2P shapes scaled down in batch only: (B,16384,3) points, 130-ch features, 128 RoIs, 512 samples."""
    B, N, C, M = 2, 16384, 130, 128
    pts, gt = scene(5, N, B)
    rng = np.random.default_rng(1)
    rois = np.ascontiguousarray(np.concatenate([gt, gt + rng.normal(0, 0.3, gt.shape).astype(np.float32)] * 4, 1)[:, :M])
    rois[:, -3:] = random_boxes(3, 3, spread=60)[None]  # a few RoIs in empty space
    feats = rng.standard_normal((B, N, C)).astype(np.float32)
    pool = RoIPointPool3d(512, width)
    with torch.no_grad():
        pooled, flag = pool(torch.from_numpy(pts).to(gpu), torch.from_numpy(feats).to(gpu), torch.from_numpy(rois).to(gpu))
    from pcdet.utils import box_utils
    big = (box_utils.enlarge_box3d(torch.from_numpy(rois).reshape(-1, 7), width) if isinstance(width, list)
           else box_utils.expand_box3d(torch.from_numpy(rois).reshape(-1, 7), width)).view(B, M, 7).numpy()
    rp, rf = oracle.roipoint_pool3d(pts, feats, big, 512)
    assert pooled.shape == (B, M, 512, 3 + C) and flag.dtype == torch.int32
    assert np.array_equal(flag.cpu().numpy(), rf) and np.array_equal(pooled.cpu().numpy(), rp)
    assert 0 < rf.sum() < B * M


@pytest.mark.parametrize("sampled", [512, 64])
def test_roipoint_pool_kernel_empty_short_and_overfull_boxes(gpu, sampled):
    """roipoint_pool_k against the oracle (roipoint_pool3d_kernel.cu:38-130) on boxes built to hold exactly 0, 1, sampled - 1, sampled,
    sampled + 1 and several thousand points: an empty box is flagged and left zero, a short list wraps around (point k % cnt), an
    overfull one keeps the FIRST `sampled` points in index order; bit for bit, features included.  Points of one box are spread over the
    whole index range (found by different waves / steps of the scan), and the two samples of the batch differ."""
    B, N, C = 2, 16384, 37
    counts = [0, 1, sampled - 1, sampled, sampled + 1, 3000, 0, 2]
    M = len(counts)
    rng = np.random.default_rng(sampled)
    pts = np.zeros((B, N, 3), np.float32)
    boxes = np.zeros((B, M, 7), np.float32)
    for b in range(B):
        pts[b] = rng.uniform(200, 300, (N, 3))                      # far from every box
        perm = rng.permutation(N)
        used = 0
        for j, cnt in enumerate(counts):
            centre = np.array([10.0 * j, 5.0 * b, 0.0], np.float32)
            boxes[b, j] = [*centre, 4.0, 2.0, 1.5, 0.3 * j]
            rows = perm[used:used + cnt]
            used += cnt
            pts[b, rows] = centre + rng.uniform(-0.3, 0.3, (cnt, 3)).astype(np.float32)   # well inside, any heading
    feats = rng.standard_normal((B, N, C)).astype(np.float32)
    pool = RoIPointPool3d(sampled, 0.0)
    with torch.no_grad():
        pooled, flag = pool(torch.from_numpy(pts).to(gpu), torch.from_numpy(feats).to(gpu), torch.from_numpy(boxes).to(gpu))
    rp, rf = oracle.roipoint_pool3d(pts, feats, boxes, sampled)
    assert np.array_equal(flag.cpu().numpy(), rf) and np.array_equal(pooled.cpu().numpy(), rp)
    assert rf.tolist() == [[int(c == 0) for c in counts]] * B
    inside = oracle.points_in_boxes_gpu(pts, boxes)
    for b in range(B):
        assert [(inside[b] == j).sum() for j in range(M)] == counts          # the construction holds what it says
    assert not pooled[:, 0].any() and not pooled[:, 6].any()
    one = pooled[0, 1].cpu().numpy()
    assert (one == one[0]).all()                                             # a single inside point fills every slot


@pytest.mark.parametrize("method", ["max", "avg"])
def test_roiaware_pool3d_forward_backward(gpu, method):
    pts, gt = scene(9, 16384, 1)
    rois = np.concatenate([gt[0], gt[0] + 0.2], 0).astype(np.float32)
    rng = np.random.default_rng(2)
    feats = rng.standard_normal((16384, 16)).astype(np.float32)
    pool = roiaware_pool3d_utils.RoIAwarePool3d(out_size=(6, 5, 4), max_pts_each_voxel=8)
    f = torch.from_numpy(feats).to(gpu).requires_grad_(True)
    out = pool(torch.from_numpy(rois).to(gpu), torch.from_numpy(pts[0]).to(gpu), f, pool_method=method)
    rp, ram, rvox = oracle.roiaware_pool3d(rois, pts[0], feats, (6, 5, 4), 8, method)
    if method == "max":
        assert np.array_equal(out.detach().cpu().numpy(), rp)
    else:
        assert np.abs(out.detach().cpu().numpy() - rp).max() < 1e-6
    g = rng.standard_normal(rp.shape).astype(np.float32)
    out.backward(torch.from_numpy(g).to(gpu))
    ref = np.zeros_like(feats)
    if method == "max":
        valid = ram >= 0
        np.add.at(ref, (ram[valid], np.broadcast_to(np.arange(16), ram.shape)[valid]), g[valid])
    else:
        R = rvox.reshape(-1, 8)
        G = g.reshape(-1, 16)
        for v in range(R.shape[0]):
            n = R[v, 0]
            if n:
                np.add.at(ref, R[v, 1:1 + n], G[v] * np.float32(1.0 / max(float(n), 1.0)))
    assert np.abs(f.grad.cpu().numpy() - ref).max() < 1e-5
    # the integer by-products through the ext-level API
    from pcdet.ops.roiaware_pool3d import roiaware_pool3d_cuda as ext
    R = rois.shape[0]
    pooled = torch.zeros((R, 6, 5, 4, 16), device=gpu)
    am = torch.zeros((R, 6, 5, 4, 16), dtype=torch.int32, device=gpu)
    vox = torch.zeros((R, 6, 5, 4, 8), dtype=torch.int32, device=gpu)
    ext.forward(torch.from_numpy(rois).to(gpu), torch.from_numpy(pts[0]).to(gpu), torch.from_numpy(feats).to(gpu), am, vox, pooled,
                {"max": 0, "avg": 1}[method])
    cnt = rvox[..., 0]
    assert np.array_equal(vox.cpu().numpy()[..., 0], cnt)
    got = vox.cpu().numpy()
    for k in range(1, 8):  # slots beyond the count are don't-care zeros in both
        sel = cnt >= k
        assert np.array_equal(got[..., k][sel], rvox[..., k][sel])
    if method == "max":
        assert np.array_equal(am.cpu().numpy(), ram)
    assert (cnt == 7).any() or cnt.max() >= 3  # some voxels hold several points (order preservation is exercised)
