"""GPU parity of the pointnet2 ops (A8-A12) through the pcdet API against the oracle restatement.

Bar: index outputs (FPS order, ball/voxel query members, 3-NN indices) bit-exact; squared distances bit-exact
(same fp32 op order, no fused multiply-add); interpolated / grouped features <= 1e-6 (copies and 3-term sums);
backward passes against torch autograd on an index-based re-expression, <= 1e-5."""
import numpy as np
import pytest
import torch

import oracle
from fv2p_harness import synth
from pcdet.ops.pointnet2.pointnet2_batch import pointnet2_modules as bmod
from pcdet.ops.pointnet2.pointnet2_batch import pointnet2_utils as bu
from pcdet.ops.pointnet2.pointnet2_stack import pointnet2_utils as su
from pcdet.ops.pointnet2.pointnet2_stack import voxel_query_utils as vq

pytestmark = pytest.mark.gpu


def T(a, gpu):
    return torch.from_numpy(np.ascontiguousarray(a)).to(gpu)


@pytest.mark.parametrize("n,m", [(16384, 16384), (16384, 4096), (5000, 1024), (700, 256), (100, 50), (20000, 512), (3, 3)])
def test_fps_order_bit_exact(gpu, n, m):
    pts = np.stack([synth.lidar_cloud(n + b, max(n, 64))[:n, :3] for b in range(2)])
    out = bu.furthest_point_sample(T(pts, gpu), m)
    ref, _ = oracle.furthest_point_sample(pts, m)
    assert out.dtype == torch.int32 and np.array_equal(out.cpu().numpy(), ref)
    assert (ref[:, 0] == 0).all()
    out2 = su.furthest_point_sample(T(pts[:1], gpu), m)  # the stack module binds the same kernel
    assert np.array_equal(out2.cpu().numpy(), ref[:1])


def test_fps_ties_follow_reference_reduction(gpu):
    """Lattice points give many exactly-equal distances: the winner must follow the reference's shared-memory tree
    (lower slot kept on ties at every halving step => smallest bit-reversed thread index wins)."""
    g = np.stack(np.meshgrid(np.arange(16), np.arange(16), np.arange(8), indexing="ij"), -1).reshape(1, -1, 3).astype(np.float32)
    out = bu.furthest_point_sample(T(g, gpu), 300)
    ref, _ = oracle.furthest_point_sample(g, 300)
    assert np.array_equal(out.cpu().numpy(), ref)


def test_fps_lattice_ties_on_the_wave_bucket_kernel(gpu):
    """Same tie rule on the register-resident wave-bucket kernel (n >= 2048, m >= 1024): a 16 x 16 x 16 lattice makes almost
    every round a many-way tie, and sampling more points than there are exhausts the cloud (all distances zero)."""
    g = np.stack(np.meshgrid(np.arange(16), np.arange(16), np.arange(16), indexing="ij"), -1).reshape(1, -1, 3).astype(np.float32)
    for m in (1500, 4096, 4500):
        out = bu.furthest_point_sample(T(g, gpu), m)
        ref, _ = oracle.furthest_point_sample(g, m)
        assert np.array_equal(out.cpu().numpy(), ref), m


def test_ball_query_group_gather_batch(gpu):
    """RoI-head shapes (iouguided_roi_head.py:276): many small samples, 512 points, 216 centres, radii 0.8 / 1.6."""
    rng = np.random.default_rng(0)
    B, N, M, C = 48, 512, 216, 128
    xyz = rng.uniform(-2, 2, (B, N, 3)).astype(np.float32)
    new = rng.uniform(-2, 2, (B, M, 3)).astype(np.float32)
    feats = rng.standard_normal((B, C, N)).astype(np.float32)
    for r, ns in [(0.8, 16), (1.6, 32), (0.05, 8)]:
        idx = bu.ball_query(r, ns, T(xyz, gpu), T(new, gpu))
        ref = oracle.ball_query_batch(r, ns, xyz, new)
        assert np.array_equal(idx.cpu().numpy(), ref)
    ft = T(feats, gpu).requires_grad_(True)
    g = bu.grouping_operation(ft, idx)
    assert np.array_equal(g.detach().cpu().numpy(), oracle.group_points_batch(feats, ref))
    go = torch.randn(g.shape, device=gpu)
    g.backward(go)
    ref_grad = torch.zeros((B, C, N), device=gpu)
    ref_grad.scatter_add_(2, idx.long().view(B, 1, -1).expand(B, C, -1), go.view(B, C, -1))
    assert (ft.grad - ref_grad).abs().max().item() < 1e-5 * ref_grad.abs().max().item()  # atomic-add order only
    # gather
    sel = T(rng.integers(0, N, (B, 64)).astype(np.int32), gpu)
    ft2 = T(feats, gpu).requires_grad_(True)
    ga = bu.gather_operation(ft2, sel)
    assert torch.equal(ga, torch.gather(ft2, 2, sel.long().unsqueeze(1).expand(B, C, 64)))
    ga.sum().backward()
    cnt = torch.zeros((B, N), device=gpu).scatter_add_(1, sel.long(), torch.ones((B, 64), device=gpu))
    assert torch.equal(ft2.grad, cnt.unsqueeze(1).expand(B, C, N))
    # QueryAndGroup + SA module run end to end on the ops
    sa = bmod.PointnetSAModuleMSG(npoint=M, radii=[0.8, 1.6], nsamples=[16, 32], mlps=[[C, 32], [C, 32]], use_xyz=True).to(gpu)
    nx, nf = sa(T(xyz, gpu), T(feats, gpu), new_xyz=T(new, gpu))
    assert nf.shape == (B, 64, M)
    nx2, nf2 = sa(T(xyz[:2], gpu), T(feats[:2], gpu))  # FPS-chosen centres
    assert nx2.shape == (2, M, 3)


@pytest.mark.parametrize("b,n,m", [(384, 512, 216), (7, 512, 217), (3, 100, 5), (2, 70, 1), (1, 1000, 129)])
@pytest.mark.parametrize("nsample", [16, 32, 5])
def test_ball_query_batch_kernel_at_the_roi_head_shapes(gpu, b, n, m, nsample):
    """ball_query_batch_k (a wave per query, 64 candidates per step, ballot-ordered stores) against oracle_ball_query_batch
    (ball_query_gpu.cu:15-51) bit for bit: the RoI head's 384 x 216 queries over 512 points with 16 / 32 samples, query counts that
    are not a multiple of the queries per workgroup, balls without a hit (index 0 in every slot: the caller's zero fill), balls with
    one hit (repeated into every slot), balls with more than 64 hits (the first `nsample` in index order, found in the first 64-candidate
    step or across steps), fewer points than a step."""
    rng = np.random.default_rng(b * 1000 + m + nsample)
    xyz = rng.uniform(-2, 2, (b, n, 3)).astype(np.float32)
    new = rng.uniform(-2, 2, (b, m, 3)).astype(np.float32)
    new[:, 0] = 50.0                                     # no point within any radius: zero hits
    xyz[:, : n // 2] *= 0.05                             # half of the points in a 0.2-wide clump: > 64 hits for centres near it
    new[:, m // 2] = 0.0
    if m > 3:
        new[:, 1] = xyz[:, n - 1] + 1e-4                 # next to the LAST point, away from the clump: few hits, found in the last step
    for radius in (0.8, 1.6, 0.01):
        ref = oracle.ball_query_batch(radius, nsample, xyz, new)
        idx = bu.ball_query(radius, nsample, T(xyz, gpu), T(new, gpu))
        assert idx.dtype == torch.int32 and tuple(idx.shape) == (b, m, nsample)
        assert np.array_equal(idx.cpu().numpy(), ref), radius
        if m > 1:
            assert not ref[:, 0].any()                             # the far query: no hit, the caller's zero fill stays
        if radius >= 0.8 and n // 2 > 64:
            d2 = ((xyz - new[:, m // 2, None, :]) ** 2).sum(-1)
            assert ((d2 < radius * radius).sum(1) > 64).all()      # the clump really is a > 64-hit ball


def test_three_nn_and_interpolate_batch(gpu):
    """V2P decoder shape: 16384 keypoints against the voxel centres of one level (top3_interpolate)."""
    rng = np.random.default_rng(1)
    kp = synth.lidar_cloud(7, 16384)[:, :3]
    centres = (np.floor(synth.lidar_cloud(8, 12000)[:, :3] / 0.4) * 0.4 + 0.2).astype(np.float32)  # lattice: many exact ties
    feats = rng.standard_normal((centres.shape[0], 64)).astype(np.float32)
    dist, idx = bu.three_nn(T(kp[None], gpu), T(centres[None], gpu))
    d2, ridx = oracle.three_nn_batch(kp[None], centres[None])
    assert np.array_equal(idx.cpu().numpy(), ridx)
    assert np.array_equal(dist.cpu().numpy(), np.sqrt(d2))
    ft = T(feats, gpu).requires_grad_(True)
    out = bu.top3_interpolate(T(centres, gpu), T(kp, gpu), ft)
    w = 1.0 / (np.sqrt(d2) + 1e-8)
    w = (w / w.sum(2, keepdims=True)).astype(np.float32)
    ref = oracle.three_interpolate_batch(feats.T[None], ridx, w)[0].T
    assert out.shape == (16384, 64) and np.abs(out.detach().cpu().numpy() - ref).max() < 1e-5
    go = torch.randn(out.shape, device=gpu)
    out.backward(go)
    wt, it = T(w[0], gpu), T(ridx[0], gpu).long()
    rg = torch.zeros_like(ft)
    for j in range(3):
        rg.index_add_(0, it[:, j], go * wt[:, j:j + 1])
    assert (ft.grad - rg).abs().max().item() < 1e-5 * rg.abs().max().item()
    # fewer than 3 known points: untouched slots keep index 0 and distance inf (interpolate_gpu.cu:37-57)
    dist, idx = bu.three_nn(T(kp[None, :100], gpu), T(centres[None, :2], gpu))
    assert torch.isinf(dist[0, :, 2]).all() and (idx[0, :, 2] == 0).all()
    xg = bu.top3_interpolate_with_grad(T(centres, gpu), T(kp[:2000], gpu), T(feats, gpu))
    assert xg.shape == (2000, 64)


def test_stack_ops(gpu):
    rng = np.random.default_rng(2)
    cnt_xyz, cnt_new = np.array([3000, 1, 2500], np.int32), np.array([300, 200, 290], np.int32)
    xyz = rng.uniform(-3, 3, (int(cnt_xyz.sum()), 3)).astype(np.float32)
    new = rng.uniform(-3, 3, (int(cnt_new.sum()), 3)).astype(np.float32)
    feats = rng.standard_normal((xyz.shape[0], 16)).astype(np.float32)
    idx, empty = su.ball_query(0.5, 16, T(xyz, gpu), T(cnt_xyz, gpu), T(new, gpu), T(cnt_new, gpu))
    raw = oracle.ball_query_stack(0.5, 16, xyz, cnt_xyz, new, cnt_new)
    rempty = raw[:, 0] == -1
    raw[rempty] = 0
    assert np.array_equal(idx.cpu().numpy(), raw) and np.array_equal(empty.cpu().numpy(), rempty)
    assert rempty.any() and not rempty.all()
    ft = T(feats, gpu).requires_grad_(True)
    g = su.grouping_operation(ft, T(cnt_xyz, gpu), idx, T(cnt_new, gpu))
    starts = np.concatenate([[0], np.cumsum(cnt_xyz)[:-1]])
    bs_of = np.repeat(np.arange(3), cnt_new)
    gidx = raw + starts[bs_of][:, None]
    assert np.array_equal(g.detach().cpu().numpy(), feats[gidx].transpose(0, 2, 1))
    go = torch.randn(g.shape, device=gpu)
    g.backward(go)
    rg = torch.zeros_like(ft).index_add_(0, T(gidx.reshape(-1), gpu).long(), go.permute(0, 2, 1).reshape(-1, 16))
    assert (ft.grad - rg).abs().max().item() < 1e-5 * rg.abs().max().item()
    nf, _ = su.QueryAndGroup(0.5, 16)(T(xyz, gpu), T(cnt_xyz, gpu), T(new, gpu), T(cnt_new, gpu), T(feats, gpu))
    assert nf.shape == (new.shape[0], 19, 16) and (nf[T(rempty, gpu)] == 0).all()
    # three_nn / interpolate (unknown = new, known = xyz)
    dist, i3 = su.three_nn(T(new, gpu), T(cnt_new, gpu), T(xyz, gpu), T(cnt_xyz, gpu))
    d2, r3 = oracle.three_nn_stack(new, cnt_new, xyz, cnt_xyz)
    assert np.array_equal(i3.cpu().numpy(), r3) and np.array_equal(dist.cpu().numpy(), np.sqrt(d2))
    w = torch.softmax(torch.randn((new.shape[0], 3), device=gpu), 1)
    ft2 = T(feats, gpu).requires_grad_(True)
    out = su.three_interpolate(ft2, i3, w)
    ref = sum(ft2.detach()[i3[:, j].long()] * w[:, j:j + 1] for j in range(3))
    assert (out - ref).abs().max().item() < 1e-5
    out.sum().backward()
    rg = torch.zeros_like(ft2)
    for j in range(3):
        rg.index_add_(0, i3[:, j].long(), w[:, j:j + 1].expand(-1, 16))
    assert (ft2.grad - rg).abs().max().item() < 1e-5 * rg.abs().max().item()


def test_voxel_query(gpu):
    rng = np.random.default_rng(3)
    Z, Y, X = 8, 40, 40
    n = 2000
    flat = rng.choice(Z * Y * X, n, replace=False)
    coords = np.stack([np.zeros(n, np.int64), flat // (Y * X), (flat // X) % Y, flat % X], 1).astype(np.int32)
    vs = np.array([0.2, 0.1, 0.1], np.float32)  # z, y, x voxel size
    xyz = (coords[:, [3, 2, 1]] * vs[[2, 1, 0]] + rng.uniform(0, 1, (n, 3)) * vs[[2, 1, 0]]).astype(np.float32)
    vol = -np.ones((1, Z, Y, X), np.int32)
    vol[0, coords[:, 1], coords[:, 2], coords[:, 3]] = np.arange(n)
    q = rng.integers(0, n, 500)
    new_coords, new_xyz = np.ascontiguousarray(coords[q]), np.ascontiguousarray(xyz[q] + 0.01)
    idx, empty = vq.voxel_query([1, 2, 2], 0.25, 16, T(xyz, gpu), T(new_xyz, gpu), T(new_coords, gpu), T(vol, gpu))
    raw = oracle.voxel_query_stack([1, 2, 2], 0.25, 16, xyz, new_xyz, new_coords, vol)
    re = raw[:, 0] == -1
    raw[re] = 0
    assert np.array_equal(idx.cpu().numpy(), raw) and np.array_equal(empty.cpu().numpy(), re)
    grp = vq.VoxelQueryAndGrouping([1, 2, 2], 0.25, 16)
    gf, gx, em = grp(T(new_coords, gpu), T(xyz, gpu), T(np.array([n], np.int32), gpu), T(new_xyz, gpu),
                     T(np.array([500], np.int32), gpu), T(rng.standard_normal((n, 8)).astype(np.float32), gpu), T(vol, gpu))
    assert gf.shape == (500, 8, 16) and gx.shape == (500, 3, 16)


def test_errors_are_exceptions_not_exit(gpu):
    with pytest.raises(Exception):
        bu.ball_query(1.0, 8, torch.zeros(1, 10, 3), torch.zeros(1, 4, 3))  # CPU tensors: the reference would exit(-1)


@pytest.mark.parametrize("n,m,b", [(16384, 2048, 2), (4099, 1500, 3), (2048, 2048, 1), (9000, 1024, 2), (9000, 64, 2), (16384, 16384, 3),
                                   (20000, 16384, 2), (24576, 4096, 1), (15000, 16384, 1), (12289, 3000, 2),
                                   (24577, 1100, 1), (40000, 4096, 2), (180000, 2048, 1), (100000, 16384, 1), (262144, 300, 1)])
def test_fps_bucketed_kernel_indices_and_running_distances(gpu, n, m, b):
    """The bucketed (lazy) FPS kernel — Morton-sorted buckets skipped when the new point cannot lower any of their
    running distances — must be indistinguishable from the reference loop: same indices AND same final `temp`
    (sampling_gpu.cu:100-216), also with duplicated points (exact ties) and n not a multiple of the bucket size."""
    from pcdet.ops.pointnet2.pointnet2_batch import pointnet2_batch_cuda as ext
    gen = synth.waymo_like_cloud if n > 30000 else synth.lidar_cloud      # n > 24576 runs the streaming kernel (points stay in memory)
    pts = np.stack([gen(3 * n + s, n)[:, :3] for s in range(b)])
    pts[:, n // 2: n // 2 + 200] = pts[:, :200]          # duplicates: zero distances and exact ties
    ref_idx, ref_temp = oracle.furthest_point_sample(pts, m)
    xyz = T(pts, gpu)
    temp = torch.full((b, n), 1e10, dtype=torch.float32, device=gpu)
    idx = torch.zeros((b, m), dtype=torch.int32, device=gpu)
    ext.furthest_point_sampling_wrapper(b, n, m, xyz, temp, idx)
    assert np.array_equal(idx.cpu().numpy(), ref_idx)
    assert np.array_equal(temp.cpu().numpy(), ref_temp)


@pytest.mark.parametrize("r,n,m,s", [(5, 100, 30, 16), (3, 512, 216, 32), (2, 640, 17, 16), (1, 64, 1, 32)])
def test_fused_grid_set_abstraction_matches_the_grouped_formulation(gpu, r, n, m, s):
    """csrc/sa_fused.hip against the grouped-tensor formulation it replaces (grouping_operation + 1x1 conv + ReLU + max over the
    samples, pointnet2_modules.py:30-62): forward 1e-5, gradients for points, centres and the weight 1e-4 relative.  The index
    lists contain repeats (ball query pads with its first hit), i.e. exact ties of the maximum."""
    from pcdet.ops.pointnet2.pointnet2_batch import fused
    g = torch.Generator().manual_seed(r * 1000 + n)
    pp = torch.randn(r, n, 64, generator=g).to(gpu).requires_grad_(True)
    pc = (torch.randn(r, m, 64, generator=g) * 0.5).to(gpu).requires_grad_(True)
    w2 = (torch.randn(64, 64, generator=g) * 0.2).to(gpu).requires_grad_(True)
    idx = torch.randint(0, n, (r, m, s), generator=g, dtype=torch.int32)
    idx[:, :, s // 2:] = idx[:, :, :1]                       # padded tail: repeats of the first hit
    idx[:, ::3, :] = idx[:, ::3, :1]                         # centres with a single distinct neighbour
    idx = idx.to(gpu)
    assert fused.supported(pp, idx)
    out = fused.sa_grid_max(pp, pc, idx, w2)
    go = torch.randn(out.shape, generator=g).to(gpu)
    out.backward(go)
    got = [t.grad.clone() for t in (pp, pc, w2)]
    for t in (pp, pc, w2):
        t.grad = None
    grouped = bu.grouping_operation(pp.transpose(1, 2).contiguous(), idx)                  # (r, 64, m, s)
    h1 = torch.relu(grouped - pc.transpose(1, 2).unsqueeze(-1))
    h2 = torch.relu(torch.einsum("oc,rcms->roms", w2, h1))
    ref = h2.amax(dim=-1).transpose(1, 2)
    ref.backward(go)
    rel = lambda a, b: float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))
    assert rel(out, ref) < 1e-5
    for a, t in zip(got, (pp, pc, w2)):
        assert rel(a, t.grad) < 1e-4


def test_fused_grid_set_abstraction_records_the_first_sample_that_attains_the_maximum(gpu):
    """fv2p_sa_grid_fwd's arg output (what backward routes the gradient by): per (centre, channel) the FIRST sample in sample order
    that attains the maximum — F.max_pool2d's index (pointnet2_modules.py:57-59) — or 255 where the maximum is 0.  Index lists with
    padded repeats (ball query pads with its first hit): the repeats are exact ties, the recorded sample must be the earliest
    slot holding the winning point."""
    import fv2p_native
    r, n, m, s = 4, 200, 37, 32
    g = torch.Generator().manual_seed(9)
    pp = torch.randn(r, n, 64, generator=g).to(gpu)
    pc = (torch.randn(r, m, 64, generator=g) * 0.5).to(gpu)
    w2 = (torch.randn(64, 64, generator=g) * 0.2).to(gpu)
    idx = torch.randint(0, n, (r, m, s), generator=g, dtype=torch.int32)
    idx[:, :, 20:] = idx[:, :, :1]                           # padded tail
    idx[:, 5, :] = idx[:, 5, :1]                             # a centre with one distinct neighbour
    idx[:, :, 7] = idx[:, :, 3]                              # a repeat in the middle
    idx = idx.to(gpu)
    out = torch.empty(r, m, 64, device=gpu)
    arg = torch.empty(r, m, 64, dtype=torch.uint8, device=gpu)
    fv2p_native.call("fv2p_sa_grid_fwd", pp, pc, idx, w2, r, n, m, s, 64, out, arg, fv2p_native.stream())
    grouped = bu.grouping_operation(pp.transpose(1, 2).contiguous(), idx)                  # (r, 64, m, s)
    h2 = torch.relu(torch.einsum("oc,rcms->roms", w2, torch.relu(grouped - pc.transpose(1, 2).unsqueeze(-1))))   # (r, 64, m, s)
    best = h2.argmax(dim=-1).transpose(1, 2)                                               # (r, m, 64): some sample holding the maximum
    a = arg.long()
    live = out > 0
    assert bool((a[~live] == 255).all()) and bool((a[live] < s).all()) and int(live.sum()) > 1000
    # the recorded sample holds the same POINT as torch's arg-max sample ...
    pt_of = lambda samp: torch.gather(idx.long().unsqueeze(2).expand(r, m, 64, s), 3, samp.clamp(max=s - 1).unsqueeze(-1)).squeeze(-1)
    same_point = pt_of(a) == pt_of(best)
    # (two different points within rounding of each other can swap between the MFMA and the einsum order: a handful at most)
    assert int((~same_point & live).sum()) <= 8, int((~same_point & live).sum())
    # ... and is the EARLIEST slot holding that point
    slots = idx.long().unsqueeze(2).expand(r, m, 64, s) == pt_of(a).unsqueeze(-1)
    first = slots.float().argmax(dim=-1)
    assert bool((first[live] == a[live]).all())


@pytest.mark.parametrize("cell", [None, 0.8, 0.1, 7.0])
def test_three_nn_grid_search_is_the_scan_bit_for_bit(gpu, cell, monkeypatch):
    """fv2p_three_nn_stack_grid against the oracle's scan: decoder shapes (16384 key points per sample against the voxel centres of a
    level: lattice, exact ties everywhere), a sample with two known points (untouched slots: index 0, inf), a sample whose queries
    lie far outside the known points' box (ring limit -> per-query scan), with and without a spacing hint, hints far too small and
    far too large.  idx and distances bit for bit."""
    rng = np.random.default_rng(4)
    knowns, unknowns = [], []
    for s, (nk, nu) in enumerate([(12000, 16384), (9000, 16384), (2, 300), (1500, 400)]):
        c = synth.lidar_cloud(20 + s, max(nk, 64))[:, :3]
        k = np.unique(np.floor(c / 0.4).astype(np.int64), axis=0)
        k = ((k + 0.5) * 0.4).astype(np.float32)[:nk]
        k = k[rng.permutation(k.shape[0])]
        u = synth.lidar_cloud(30 + s, nu)[:, :3]
        if s == 3:
            u = u + np.float32([300.0, -200.0, 40.0])          # far from every known point
        if s == 1:
            u[:500] = k[rng.integers(0, k.shape[0], 500)] - np.float32(0.2)    # cell corners: up to 8 equidistant centres
        knowns.append(k); unknowns.append(u.astype(np.float32))
    kc, uc = np.array([k.shape[0] for k in knowns], np.int32), np.array([u.shape[0] for u in unknowns], np.int32)
    known, unknown = np.concatenate(knowns), np.concatenate(unknowns)
    monkeypatch.setattr(su, "GRID_MIN_KNOWN", 0)              # the grid path is the one under test, whatever the size threshold
    dist, idx = su.three_nn(T(unknown, gpu), T(uc, gpu), T(known, gpu), T(kc, gpu), cell)
    d2, ridx = oracle.three_nn_stack(unknown, uc, known, kc)
    assert np.array_equal(idx.cpu().numpy(), ridx)
    assert np.array_equal(dist.cpu().numpy(), np.sqrt(d2))
    lo = int(uc[:2].sum())
    assert np.isinf(np.sqrt(d2)[lo:lo + 300, 2]).all()          # the two-point sample really has untouched slots


@pytest.mark.parametrize("stride", [1, 4])
def test_three_nn_grid_is_the_scan_at_waymo_size(gpu, stride, monkeypatch):
    """BASELINE configs[4] size (the oracle's scan would take minutes): the hashed-grid search against this library's own scan, which
    the tests above pin to the oracle — 2 x 16 384 queries (the clouds' own points plus 2 000 scattered far and wide: sparse regions
    are where the second pass with its shells and its per-wave scan runs) against the voxel centres of a Waymo-shaped level
    (~57 k per sample at stride 1), with the decoder's spacing hint.  idx and distances bit for bit."""
    import os
    vsz = np.float32([0.1, 0.1, 0.15]) * stride
    lo = np.float32([-75.2, -75.2, -2.0])
    rng = np.random.default_rng(8)
    knowns, unknowns = [], []
    for s in range(2):
        c = synth.waymo_like_cloud(40 + s, 180000)[:, :3]
        cell = np.unique(np.floor((c - lo) / vsz).astype(np.int64), axis=0)
        knowns.append(((cell + 0.5) * vsz + lo).astype(np.float32))
        u = c[rng.permutation(c.shape[0])[:16384]].copy()
        u[:2000] = rng.uniform([-90, -90, -4], [90, 90, 6], (2000, 3)).astype(np.float32)
        unknowns.append(u.astype(np.float32))
    kc, uc = np.array([k.shape[0] for k in knowns], np.int32), np.array([u.shape[0] for u in unknowns], np.int32)
    known, unknown = T(np.concatenate(knowns), gpu), T(np.concatenate(unknowns), gpu)
    monkeypatch.setenv("FV2P_NN_GRID", "0")
    d0, i0 = su.three_nn(unknown, T(uc, gpu), known, T(kc, gpu))
    monkeypatch.setenv("FV2P_NN_GRID", "1")
    monkeypatch.setattr(su, "GRID_MIN_KNOWN", 0)
    d1, i1 = su.three_nn(unknown, T(uc, gpu), known, T(kc, gpu), 2.0 * float(vsz[0]))
    assert int(kc.min()) > (50000 if stride == 1 else 10000)
    assert torch.equal(i0, i1) and torch.equal(d0, d1)


@pytest.mark.parametrize("n,m,c", [(49152, 35146, 16), (49152, 5186, 128), (32768, 200000, 32), (9000, 700, 64), (9001, 333, 67), (10, 5, 4)])
def test_interpolation_gradient_gather_form_equals_the_scatter_form(gpu, n, m, c, monkeypatch):
    """fv2p_three_interpolate_stack_grad_gather (sorted (row, entry) keys summed in segments; no float atomics, no zero fill) against the
    oracle's float64 accumulation and against the scatter form, at the decoder's shapes: rows nobody reads come out exactly zero, rows
    read by thousands of queries (a skewed idx: runs that cross many segments) agree to 1e-5 relative, two calls agree bit for bit; through
    the autograd op it is the default from 8192 queries on, FV2P_INTERP_GATHER=0 selects the scatter form."""
    import fv2p_native
    rng = np.random.default_rng(n + c)
    idx = rng.integers(0, m, (n, 3)).astype(np.int32)
    idx[: n // 4] = rng.integers(0, max(m // 50, 1), (n // 4, 3))          # hot rows: long entry lists
    idx[:, 2][rng.random(n) < 0.1] = idx[:, 0][rng.random(n) < 0.1][0] if n > 10 else idx[0, 0]
    unread = np.setdiff1d(np.arange(m), idx.reshape(-1))
    w = rng.random((n, 3)).astype(np.float32)
    g = rng.standard_normal((n, c)).astype(np.float32)
    want = np.zeros((m, c), np.float64)
    for k in range(3):
        np.add.at(want, idx[:, k], g.astype(np.float64) * w[:, k:k + 1])
    tg, ti, tw = T(g, gpu), T(idx, gpu), T(w, gpu)
    out = torch.full((m, c), 7.0, device=gpu)                              # no zero fill expected of the caller
    ws = torch.empty(int(fv2p_native.lib().fv2p_three_interpolate_stack_grad_ws_bytes(n, c, m)), dtype=torch.uint8, device=gpu)
    fv2p_native.call("fv2p_three_interpolate_stack_grad_gather", n, c, m, tg, ti, tw, out, ws, ws.numel(), fv2p_native.stream())
    got = out.cpu().numpy()
    out2 = torch.full((m, c), -3.0, device=gpu)
    fv2p_native.call("fv2p_three_interpolate_stack_grad_gather", n, c, m, tg, ti, tw, out2, ws, ws.numel(), fv2p_native.stream())
    assert torch.equal(out, out2)                                          # fixed association: bit-identical from call to call
    scale = np.abs(want).max()
    assert np.abs(got - want).max() < 1e-5 * scale
    if unread.size:
        assert not got[unread].any()
    feats = torch.randn(m, c, device=gpu, requires_grad=True)
    monkeypatch.setenv("FV2P_INTERP_GATHER", "1")
    su.three_interpolate(feats, ti, tw).backward(tg)                       # the autograd op: gather form from 8192 queries on
    assert np.abs(feats.grad.cpu().numpy() - want).max() < 1e-5 * scale
    monkeypatch.setenv("FV2P_INTERP_GATHER", "0")
    feats.grad = None
    su.three_interpolate(feats, ti, tw).backward(tg)
    assert np.abs(feats.grad.cpu().numpy() - want).max() < 1e-5 * scale


@pytest.mark.parametrize("n,m,c", [(8192, 3000, 64), (20000, 4000, 128), (9001, 333, 20)])
def test_default_interpolation_gradient_through_autograd_and_from_unaligned_buffers(gpu, n, m, c):
    """The autograd op's DEFAULT gradient from 8192 queries on (no environment override: the segmented gather form,
    pointnet2_utils.py _interp_grad) against a float64 accumulation, 1e-5 of the gradient's scale; and the same entry point fed
    buffers that start 4 bytes past a 16-byte boundary (a tensor view with a storage offset, a C caller's pointer): the kernels
    fall back to scalar accesses instead of faulting on a misaligned 16-byte access."""
    import os
    import fv2p_native
    assert "FV2P_INTERP_GATHER" not in os.environ and n >= su.GATHER_GRAD_MIN_QUERIES
    rng = np.random.default_rng(n + m)
    idx = rng.integers(0, m, (n, 3)).astype(np.int32)
    idx[: n // 8] = rng.integers(0, 4, (n // 8, 3))                       # four hot rows
    w = rng.random((n, 3)).astype(np.float32)
    g = rng.standard_normal((n, c)).astype(np.float32)
    want = np.zeros((m, c), np.float64)
    for k in range(3):
        np.add.at(want, idx[:, k], g.astype(np.float64) * w[:, k:k + 1])
    scale = np.abs(want).max()
    feats = torch.randn(m, c, device=gpu, requires_grad=True)
    su.three_interpolate(feats, T(idx, gpu), T(w, gpu)).backward(T(g, gpu))
    assert np.abs(feats.grad.cpu().numpy() - want).max() < 1e-5 * scale
    # unaligned: both float buffers one element into their storage
    g_store = torch.empty(n * c + 1, device=gpu)
    g_view = g_store[1:].view(n, c)
    g_view.copy_(T(g, gpu))
    out_store = torch.full((m * c + 1,), 9.0, device=gpu)
    out_view = out_store[1:].view(m, c)
    assert g_view.data_ptr() % 16 == 4 and out_view.data_ptr() % 16 == 4
    ws = torch.empty(int(fv2p_native.lib().fv2p_three_interpolate_stack_grad_ws_bytes(n, c, m)), dtype=torch.uint8, device=gpu)
    fv2p_native.call("fv2p_three_interpolate_stack_grad_gather", n, c, m, g_view, T(idx, gpu), T(w, gpu), out_view, ws, ws.numel(), fv2p_native.stream())
    torch.cuda.synchronize()
    assert float(out_store[0]) == 9.0
    if c % 4 == 0:
        assert torch.equal(out_view, feats.grad)      # same association as the aligned call: bit-identical
    else:
        assert np.abs(out_view.cpu().numpy() - want).max() < 1e-5 * scale
