"""Size-independent properties at BASELINE.json's FULL sizes, where the CPU oracle is too slow to answer (it is used at these sizes only
where it finishes in seconds): adjointness of the three conv kernels (forward, backward data, weight gradient are one bilinear form),
linearity, rulebook symmetry, voxeliser conservation laws, sampler monotonicity, NMS idempotence and separation, pooled points inside
their boxes, DCN linearity.  Each property is something the reference's ops satisfy by construction (cited), and together they tie
the kernels to each other at the sizes bench.py runs them at."""
import numpy as np
import pytest
import torch

import pcdet.ops.spconv as spconv
from boxes_util import random_boxes
from fv2p_harness import synth
from pcdet.datasets.processor.voxel_generator import points_to_voxel
from pcdet.ops.iou3d_nms import iou3d_nms_utils
from pcdet.ops.pointnet2.pointnet2_batch import pointnet2_utils as bu
from pcdet.ops.roipoint_pool3d.roipoint_pool3d_utils import RoIPointPool3d
from pcdet.ops.roiaware_pool3d import roiaware_pool3d_utils
from pcdet.ops.spconv import ops

pytestmark = pytest.mark.gpu


def kitti_batch(gpu, seeds, n_points=16384):
    feats, coords = [], []
    for b, s in enumerate(seeds):
        pts = torch.from_numpy(synth.lidar_cloud(s, n_points)).to(gpu)
        v, c, n = points_to_voxel(pts, synth.KITTI_VOXEL, synth.KITTI_RANGE, 5, True, 16000)
        feats.append(v.sum(1) / n.clamp(min=1).view(-1, 1).float())
        coords.append(torch.cat([torch.full((c.shape[0], 1), b, dtype=torch.int32, device=gpu), c], 1))
    return torch.cat(feats).contiguous(), torch.cat(coords).contiguous()


def dot(a, b):
    return float((a.double() * b.double()).sum())


@pytest.mark.parametrize("n_points,vox,rng_,mv", [(16384, "KITTI", "KITTI", 16000), (180000, "WAYMO", "WAYMO", 80000)])
def test_voxeliser_conservation_laws_at_full_size(gpu, n_points, vox, rng_, mv):
    """voxel_generator.py:136-207 at BASELINE configs[0] / [4] size: coordinates are distinct and inside the grid; every kept point
    sits in the voxel its coordinates name; the point slots of all voxels hold exactly the first <= max_points in-range points of each
    voxel in input order (checksum of checksums: per-voxel sums of the payload equal the per-voxel sums over the input); slots past
    num_points are zero."""
    pts = synth.lidar_cloud(7, n_points) if vox == "KITTI" else synth.waymo_like_cloud(7, n_points)
    vsz = np.asarray(getattr(synth, vox + "_VOXEL"), np.float32)
    rng = np.asarray(getattr(synth, rng_ + "_RANGE"), np.float32)
    d = torch.from_numpy(pts).to(gpu)
    v, c, k = points_to_voxel(d, vsz, rng, 5, True, mv)
    v, c, k = v.cpu().numpy(), c.cpu().numpy().astype(np.int64), k.cpu().numpy()
    grid = np.round((rng[3:] - rng[:3]) / vsz).astype(np.int64)
    assert (c >= 0).all() and (c[:, 0] < grid[2]).all() and (c[:, 1] < grid[1]).all() and (c[:, 2] < grid[0]).all()
    flat = (c[:, 0] * grid[1] + c[:, 1]) * grid[0] + c[:, 2]
    assert np.unique(flat).size == flat.size
    assert (k >= 1).all() and (k <= 5).all()
    slot = np.arange(5)[None, :] < k[:, None]
    assert not v[~slot].any()
    # every stored point lies in its voxel (fp32 floor((p - lo) / vs), :188)
    pc = np.floor((v[..., :3] - rng[:3]) / vsz).astype(np.int64)[slot]
    assert np.array_equal(pc[:, ::-1], np.repeat(c, k, axis=0))
    # the input, voxelised sequentially on the host by its own rule, fills the same slots: compare per-voxel payload sums
    ic = np.floor((pts[:, :3] - rng[:3]) / vsz)
    ok = ((ic >= 0) & (ic < grid[None, :])).all(1)
    iflat = (ic[:, 2].astype(np.int64) * grid[1] + ic[:, 1].astype(np.int64)) * grid[0] + ic[:, 0].astype(np.int64)
    order = {f: i for i, f in enumerate(flat.tolist())}
    want = np.zeros((flat.size, pts.shape[1]), np.float64)
    seen = np.zeros(flat.size, np.int64)
    first_new = {}
    for i in np.nonzero(ok)[0]:
        j = order.get(int(iflat[i]))
        if j is None:
            first_new.setdefault("i", i)     # a voxel beyond max_voxels: the scan stopped here (:198-199)
            break
        if seen[j] < 5:
            want[j] += pts[i]
            seen[j] += 1
    assert np.array_equal(seen, k)
    assert np.allclose(v.astype(np.float64).sum(1), want, rtol=0, atol=1e-3)


def test_submanifold_rulebook_is_symmetric_at_configs1_size(gpu):
    """getIndicePairsSubM (geometry.h:247-297): offset k pairs (i -> o) exactly when offset K-1-k pairs (o -> i); the centre offset is
    the identity; counts are symmetric.  Batch 4 of 16 384-point clouds on the full KITTI grid."""
    _, coords = kitti_batch(gpu, [3, 4, 5, 6])
    rb = ops.build_rulebook(coords, 4, [41, 1600, 1408], 3, 1, 1, 1, 0, True)
    pairs, num = rb.indice_pairs.cpu().numpy(), rb.indice_pair_num.cpu().numpy()
    n = coords.shape[0]
    assert num[13] == n and np.array_equal(np.sort(pairs[13, 0, :n]), np.arange(n)) and np.array_equal(pairs[13, 0, :n], pairs[13, 1, :n])
    assert np.array_equal(num, num[::-1])
    for k in range(13):
        a = pairs[k, :, :num[k]]
        b = pairs[26 - k, :, :num[26 - k]]
        ka = np.sort(a[0].astype(np.int64) * n + a[1])
        kb = np.sort(b[1].astype(np.int64) * n + b[0])
        assert np.array_equal(ka, kb), k


@pytest.mark.parametrize("cin,cout,stride", [(16, 16, 1), (32, 32, 1), (64, 64, 1), (128, 128, 1), (16, 32, 2), (64, 128, 2)])
def test_conv_kernels_are_one_bilinear_form_at_full_size(gpu, cin, cout, stride):
    """indiceConv / indiceConvBackward (spconv_ops.h:260-457) compute y = A(W) x, dx = A(W)^T g and dW = d/dW <A(W) x, g>, so for any
    x, g, W, W':  <conv_W(x), g> = <x, dx_W(g)>  and  <dW(x, g), W'> = <conv_W'(x), g>  - without any oracle, at the row counts of
    BASELINE configs[1] (batch 4: ~47 k rows at level 1).  Plus linearity of the forward kernel in x.  1e-4 relative (fp32 dot products
    over 1e6 ... 1e7 terms, accumulated here in float64)."""
    feats, coords = kitti_batch(gpu, [3, 4, 5, 6])
    n = coords.shape[0]
    g_ = torch.Generator(device=gpu).manual_seed(cin * 131 + cout)
    x = torch.randn(n, cin, device=gpu, generator=g_)
    x2 = torch.randn(n, cin, device=gpu, generator=g_)
    if stride == 1:
        conv = spconv.SubMConv3d(cin, cout, 3, padding=1, bias=False, indice_key="p").to(gpu)
    else:
        conv = spconv.SparseConv3d(cin, cout, 3, stride=2, padding=1, bias=False, indice_key="p").to(gpu)
    w2 = torch.randn_like(conv.weight)

    def run(inp, weight=None):
        if weight is not None:
            saved = conv.weight.data
            conv.weight.data = weight
        t = spconv.SparseConvTensor(inp, coords, [41, 1600, 1408], 4)
        out = conv(t).features
        if weight is not None:
            conv.weight.data = saved
        return out

    xr = x.clone().requires_grad_(True)
    y = run(xr)
    g = torch.randn(y.shape, device=gpu, generator=g_)
    y.backward(g)
    lhs = dot(y.detach(), g)
    scale = float(y.detach().double().norm() * g.double().norm())
    assert abs(lhs - dot(x, xr.grad)) <= 1e-4 * scale                          # backward data is the adjoint of the forward kernel
    with torch.no_grad():
        y_w2 = run(x, w2)
        assert abs(dot(conv.weight.grad, w2) - dot(y_w2, g)) <= 1e-4 * float(y_w2.double().norm() * g.double().norm())   # ... and so is the weight gradient
        y2 = run(x2)
        lin = run(0.5 * x - 2.0 * x2)
        assert float((lin - (0.5 * y.detach() - 2.0 * y2)).abs().max()) <= 1e-4 * float(y.detach().abs().max() + 2 * y2.abs().max())


def test_fps_distances_fall_and_picks_are_distinct_at_full_size(gpu):
    """furthest_point_sampling_kernel (sampling_gpu.cu:100-216), three 16 384-point clouds, all 16 384 picks: the first pick is point 0,
    every point is picked exactly once, and the distance of a pick to the set chosen before it never grows (recomputed on the host in
    float64 for every 64th round: the maximin distance sequence of farthest point sampling is non-increasing)."""
    pts = np.stack([synth.lidar_cloud(s, 16384)[:, :3] for s in (1, 2, 3)])
    idx = bu.furthest_point_sample(torch.from_numpy(pts).to(gpu), 16384).cpu().numpy()
    assert (idx[:, 0] == 0).all()
    for b in range(3):
        assert np.array_equal(np.sort(idx[b]), np.arange(16384))
        p = pts[b].astype(np.float64)
        mind = np.full(16384, np.inf)
        last, prev = idx[b, 0], np.inf
        for j in range(1, 2049):
            mind = np.minimum(mind, ((p - p[last]) ** 2).sum(1))
            last = idx[b, j]
            d = mind[last]
            assert d <= prev * (1 + 1e-6) and d >= mind.max() * (1 - 1e-5), (b, j)   # the pick attains the current maximum
            prev = d


def test_nms_survivors_are_separated_and_stable_at_full_size(gpu):
    """nms_gpu (iou3d_nms.cpp:90-136) on 9 000 boxes at the train threshold: no two survivors overlap by more than the threshold, every
    suppressed box overlaps an earlier-scored survivor by more than it, and NMS of the survivors keeps all of them."""
    boxes = torch.from_numpy(random_boxes(3, 9000)).to(gpu)
    scores = torch.rand(9000, device=gpu, generator=torch.Generator(device=gpu).manual_seed(0))
    keep, _ = iou3d_nms_utils.nms_gpu(boxes, scores, 0.8)
    keep = keep.long()
    assert 0 < keep.numel() < 9000 and torch.equal(scores[keep], scores[keep].sort(descending=True)[0])
    iou = iou3d_nms_utils.boxes_iou_bev(boxes[keep], boxes[keep])
    iou.fill_diagonal_(0)
    assert float(iou.max()) <= 0.8
    rest = torch.ones(9000, dtype=torch.bool, device=gpu)
    rest[keep] = False
    cross = iou3d_nms_utils.boxes_iou_bev(boxes[rest], boxes[keep])
    earlier = scores[keep][None, :] > scores[rest][:, None]
    assert bool(((cross > 0.8) & earlier).any(1).all())
    again, _ = iou3d_nms_utils.nms_gpu(boxes[keep], scores[keep], 0.8)
    assert again.numel() == keep.numel()


def test_pooled_points_lie_inside_their_enlarged_boxes_at_full_size(gpu):
    """roipool3d (roipoint_pool3d_kernel.cu:38-130) at FV2P's shapes (3 x 16 384 points, 130 features, 128 RoIs, 512 samples): every
    pooled point of a non-empty box is one of the cloud's points inside the enlarged box, carries that point's features, and empty boxes
    are flagged exactly when no point is inside."""
    pts, gts = [], []
    for s in range(3):
        p, bx = synth.lidar_cloud(20 + s, 16384, return_boxes=True)
        pts.append(p[:, :3])
        gts.append(bx[:16])
    pts, gts = np.stack(pts).astype(np.float32), np.stack(gts).astype(np.float32)
    rng = np.random.default_rng(1)
    rois = np.concatenate([gts + rng.normal(0, 0.2, gts.shape).astype(np.float32) for _ in range(8)], 1)
    feats = rng.standard_normal((3, 16384, 130)).astype(np.float32)
    pool = RoIPointPool3d(512, [1.0, 1.0, 1.0])
    with torch.no_grad():
        pooled, flag = pool(torch.from_numpy(pts).to(gpu), torch.from_numpy(feats).to(gpu), torch.from_numpy(rois).to(gpu))
    big = torch.from_numpy(rois).to(gpu).clone()
    big[..., 3:6] += 1.0
    for b in range(3):
        inside = roiaware_pool3d_utils.points_in_boxes_gpu(torch.from_numpy(pts[b:b + 1]).to(gpu).expand(128, -1, -1).contiguous(),
                                                          big[b].unsqueeze(1).contiguous())           # (128, 16384): 0 inside box r, -1 outside
        cnt = (inside == 0).sum(1)
        assert torch.equal(flag[b].bool(), cnt == 0)
        for r in torch.nonzero(cnt > 0).flatten().tolist()[:24]:
            rows = torch.nonzero(inside[r] == 0).flatten()[:512]
            want_xyz = torch.from_numpy(pts[b]).to(gpu)[rows]
            k = rows.numel()
            assert torch.equal(pooled[b, r, :k, :3], want_xyz)                        # the first inside points in index order
            assert torch.equal(pooled[b, r, :k, 3:], torch.from_numpy(feats[b]).to(gpu)[rows])
            if k < 512:
                assert torch.equal(pooled[b, r, k:, :3], want_xyz[torch.arange(k, 512, device=gpu) % k])   # wrapped around


def test_dcn_is_linear_in_input_and_weight_at_full_map_size(gpu):
    """modulated_deform_conv (modulated_deform_conv_cuda.cu:19-120) at the MGAF BEV size [2,128,200,176]: for fixed offsets and masks the
    output is linear in x and in the weight, and its input gradient is the adjoint of the forward map."""
    from pcdet.ops.DeformableConvolutionV2PyTorch.modules.modulated_deform_conv import ModulatedDeformConv
    torch.manual_seed(0)
    m = ModulatedDeformConv(128, 128, 3, stride=1, padding=1, deformable_groups=1, bias=False).to(gpu)
    with torch.no_grad():
        m.bias.zero_()
    x, x2 = torch.randn(2, 128, 200, 176, device=gpu), torch.randn(2, 128, 200, 176, device=gpu)
    off = torch.randn(2, 18, 200, 176, device=gpu)
    msk = torch.sigmoid(torch.randn(2, 9, 200, 176, device=gpu))
    xr = x.clone().requires_grad_(True)
    y = m(xr, off, msk)
    g = torch.randn_like(y)
    y.backward(g)
    assert abs(dot(y.detach(), g) - dot(x, xr.grad)) <= 1e-4 * float(y.detach().double().norm() * g.double().norm())
    with torch.no_grad():
        y2 = m(x2, off, msk)
        lin = m(0.5 * x - 2.0 * x2, off, msk)
        assert float((lin - (0.5 * y.detach() - 2.0 * y2)).abs().max()) <= 1e-4 * float(y.detach().abs().max() + 2 * y2.abs().max())
        w2 = torch.randn_like(m.weight)
        saved = m.weight.data
        m.weight.data = w2
        y_w2 = m(x, off, msk)
        m.weight.data = saved
    assert abs(dot(m.weight.grad, w2) - dot(y_w2, g)) <= 1e-4 * float(y_w2.double().norm() * g.double().norm())


def test_three_nn_and_interpolation_at_decoder_size(gpu):
    """three_nn / three_interpolate (interpolate_gpu.cu:16-149, stack form) at the decoder's shapes (3 x 16 384 key points against
    the ~35 k voxel centres of level 1): distances ascending and no unlisted centre of a sample closer than the third (checked against a
    blocked brute force for 2 048 random queries); the interpolation gradient - the segmented gather form, default at this size - is the
    adjoint of the forward op: <interp(f), g> = <f, grad(g)> to 1e-5."""
    from pcdet.ops.pointnet2.pointnet2_stack import pointnet2_utils as su
    _, coords = kitti_batch(gpu, [3, 4, 5])
    cnt_k = torch.bincount(coords[:, 0].long(), minlength=3).int()
    vs, lo = torch.tensor(synth.KITTI_VOXEL, device=gpu), torch.tensor(synth.KITTI_RANGE[:3], device=gpu)
    known = ((coords[:, [3, 2, 1]].float() + 0.5) * vs + lo).contiguous()
    unknown = torch.cat([torch.from_numpy(synth.lidar_cloud(30 + s, 16384)[:, :3]).to(gpu) for s in range(3)]).contiguous()
    cnt_u = torch.full((3,), 16384, dtype=torch.int32, device=gpu)
    dist, idx = su.three_nn(unknown, cnt_u, known, cnt_k)
    assert bool((dist[:, 0] <= dist[:, 1]).all()) and bool((dist[:, 1] <= dist[:, 2]).all())
    starts = torch.cumsum(cnt_k, 0) - cnt_k
    pick = torch.randperm(unknown.shape[0], device=gpu, generator=torch.Generator(device=gpu).manual_seed(0))[:2048]
    for q in pick.tolist()[:2048:8]:
        b = q // 16384
        seg = known[int(starts[b]):int(starts[b]) + int(cnt_k[b])]
        d2 = ((seg - unknown[q]) ** 2).sum(1)
        best = torch.topk(d2, 3, largest=False)[0].sqrt()
        assert torch.allclose(best, dist[q], rtol=1e-5, atol=1e-6)
        assert bool((idx[q] >= int(starts[b])).all()) and bool((idx[q] < int(starts[b]) + int(cnt_k[b])).all())
    w = 1.0 / (dist + 1e-8)
    w = (w / w.sum(1, keepdim=True)).contiguous()
    f = torch.randn(known.shape[0], 64, device=gpu, requires_grad=True)
    out = su.three_interpolate(f, idx, w)
    g = torch.randn_like(out)
    out.backward(g)
    assert unknown.shape[0] >= su.GATHER_GRAD_MIN_QUERIES
    assert abs(dot(out.detach(), g) - dot(f.detach(), f.grad)) <= 1e-5 * float(out.detach().double().norm() * g.double().norm())


def test_dense_scatter_and_bev_gather_are_adjoint_pairs_at_full_size(gpu):
    """SparseConvTensor.dense() (structure.py:57-66) at HeightCompression's size ([B,128,2,200,176]): the dense tensor holds exactly the
    rows at their cells and zeros elsewhere, and its gradient is the gather at those cells.  BEVGridPooling's bilinear gather
    (bev_grid_pooling.py:11-45) on the [3,256,200,176] map: the map gradient is the adjoint of the gather."""
    from pcdet.models.backbones_3d.pfe import bev_grid_pooling as bgp
    rng = np.random.default_rng(0)
    cells = rng.choice(3 * 2 * 200 * 176, 30000, replace=False)
    b, r = cells // (2 * 200 * 176), cells % (2 * 200 * 176)
    ind = torch.from_numpy(np.stack([b, r // (200 * 176), (r // 176) % 200, r % 176], 1).astype(np.int32)).to(gpu)
    f = torch.randn(30000, 128, device=gpu, requires_grad=True)
    d = spconv.SparseConvTensor(f, ind, [2, 200, 176], 3).dense()
    assert tuple(d.shape) == (3, 128, 2, 200, 176)
    li = ind.long()
    assert torch.equal(d[li[:, 0], :, li[:, 1], li[:, 2], li[:, 3]], f.detach())
    assert abs(float(d.detach().double().abs().sum()) - float(f.detach().double().abs().sum())) <= 1e-6 * float(f.detach().double().abs().sum())
    g = torch.randn_like(d)
    d.backward(g)
    assert torch.equal(f.grad, g[li[:, 0], :, li[:, 1], li[:, 2], li[:, 3]])
    bev = torch.randn(3, 256, 200, 176, device=gpu, requires_grad=True)
    kp = torch.rand(3, 4608, 3, device=gpu) * torch.tensor([70.4, 80.0, 4.0], device=gpu) + torch.tensor([0.0, -40.0, -3.0], device=gpu)
    out = bgp.interpolate_from_bev_features(kp, bev, 3, 8, synth.KITTI_RANGE, synth.KITTI_VOXEL)
    go = torch.randn_like(out)
    out.backward(go)
    with torch.no_grad():
        probe = torch.randn_like(bev)
        lhs = dot(bgp.interpolate_from_bev_features(kp, probe, 3, 8, synth.KITTI_RANGE, synth.KITTI_VOXEL), go)
    assert abs(lhs - dot(probe, bev.grad)) <= 1e-4 * float(probe.double().norm() * bev.grad.double().norm())
