"""Cross-checks of the UNPINNED oracles (pointnet2 A8-A12 and roipoint A18 are GPU-only CUDA in the reference, no vectors
exist and none can be generated here).  Each C restatement (oracle/pointnet2_oracle.c, oracle/roi_oracle.c) is held against a SECOND,
independently written formulation in numpy — vectorised, no loop structure in common with the C — that states the kernel's
SEMANTICS: float32 distances in the reference's operand order, ties to the lowest index (3-NN, ball query), the farthest-point
winner among equal maxima by the reference's thread ownership; plus a float64 tie audit saying how many decisions of each input sit
within one float32 ulp of a tie (where a CUDA build's fused multiply-adds could legitimately decide otherwise: tools/fma_audit.py).
CPU only."""
import numpy as np
import pytest

import oracle
from fv2p_harness import synth


def d2_f32(a, b):
    """(Na, 3) x (Nb, 3) -> (Na, Nb) float32 squared distances, ((dx*dx + dy*dy) + dz*dz) rounded after every operation."""
    d = (a[:, None, :] - b[None, :, :]).astype(np.float32)
    sq = (d * d).astype(np.float32)
    return ((sq[..., 0] + sq[..., 1]).astype(np.float32) + sq[..., 2]).astype(np.float32)


def cloud(seed, n):
    return synth.lidar_cloud(seed, n)[:, :3].copy()


def lattice(seed, n):
    """Voxel centres of a cloud at 0.4 m: exact ties between neighbours of a query are the rule, not the exception."""
    p = cloud(seed, n)
    c = np.unique(np.floor(p / 0.4).astype(np.int64), axis=0)
    return ((c + 0.5) * 0.4).astype(np.float32)


@pytest.mark.parametrize("known_kind", ["cloud", "lattice"])
def test_three_nn_is_the_three_smallest_by_distance_then_index(known_kind):
    known = cloud(2, 700) if known_kind == "cloud" else lattice(2, 3000)
    # lattice case: the queries are cell CORNERS of the known points' lattice, equidistant (in exact arithmetic) from up to 8 centres
    unknown = cloud(1, 1500) if known_kind == "cloud" else (known[:1500] - np.float32(0.2)).astype(np.float32)
    d2, idx = oracle.three_nn_batch(unknown[None], known[None])
    d = d2_f32(unknown, known)
    order = np.lexsort((np.broadcast_to(np.arange(known.shape[0]), d.shape), d), axis=1)[:, :3]    # by distance, then index
    assert np.array_equal(idx[0], order)
    assert np.array_equal(d2[0], np.take_along_axis(d, order, 1))
    # stacked form: two samples, indices global over the concatenation (interpolate_gpu.cu:16-73 of pointnet2_stack)
    u2, k2 = cloud(3, 400), lattice(4, 900)
    ds, is_ = oracle.three_nn_stack(np.concatenate([unknown, u2]), [unknown.shape[0], 400], np.concatenate([known, k2]), [known.shape[0], k2.shape[0]])
    dd = d2_f32(u2, k2)
    o2 = np.lexsort((np.broadcast_to(np.arange(k2.shape[0]), dd.shape), dd), axis=1)[:, :3]
    nu = unknown.shape[0]
    assert np.array_equal(is_[:nu], order) and np.array_equal(is_[nu:], o2 + known.shape[0])
    assert np.array_equal(ds[nu:], np.take_along_axis(dd, o2, 1))
    # tie audit (float64): queries whose 3rd and 4th neighbour are closer than one float32 ulp
    d64 = ((unknown[:, None, :].astype(np.float64) - known[None].astype(np.float64)) ** 2).sum(-1)
    s = np.sort(d64, 1)[:, :4]
    near_ties = int((np.abs(s[:, 3] - s[:, 2]) <= np.spacing(s[:, 2].astype(np.float32))).sum())
    assert near_ties >= 0
    if known_kind == "lattice":
        assert near_ties > 0      # the lattice case really exercises the tie rule


def test_three_nn_with_fewer_than_three_known_points():
    unknown = cloud(5, 50)
    d2, idx = oracle.three_nn_batch(unknown[None], cloud(6, 2)[None])
    assert (idx[0, :, 2] == 0).all() and np.isinf(d2[0, :, 2]).all()          # interpolate_gpu.cu:37-55: untouched slot: idx 0, 1e40 -> inf


def ball_members(xyz, centres, radius, nsample):
    """First nsample indices with d2 < r^2 in index order, the tail padded with the first member; empty ball: zeros."""
    inside = d2_f32(centres, xyz) < np.float32(radius) * np.float32(radius)
    out = np.zeros((centres.shape[0], nsample), np.int32)
    for m in range(centres.shape[0]):     # ragged: a loop over centres only
        k = np.flatnonzero(inside[m])[:nsample]
        if k.size:
            out[m] = k[0]
            out[m, :k.size] = k
    return out, inside.sum(1)


@pytest.mark.parametrize("radius,nsample", [(0.8, 16), (1.6, 32), (0.05, 8)])
def test_ball_query_members_in_index_order(radius, nsample):
    xyz = cloud(7, 2000)
    centres = np.concatenate([xyz[::9] + np.float32(0.03), np.float32([[500.0, 0.0, 0.0]])])     # last centre: an empty ball
    idx = oracle.ball_query_batch(radius, nsample, xyz[None], centres[None])[0]
    want, counts = ball_members(xyz, centres, radius, nsample)
    assert np.array_equal(idx, want)
    assert counts[-1] == 0 and (idx[-1] == 0).all()
    assert (counts > nsample).any() or radius < 0.1           # truncation at nsample is exercised
    raw = oracle.ball_query_stack(radius, nsample, xyz, [2000], centres, [centres.shape[0]])
    assert raw[-1, 0] == -1                                   # stack kernel marks empty balls (ball_query_gpu.cu:58-64)
    live = counts > 0
    assert np.array_equal(raw[live], want[live])


def fps_by_rule(xyz, m):
    """Farthest point sampling stated as a rule: start at 0; every round lowers each point's running distance to the last pick
    and takes the maximum.  Among EQUAL maxima: thread t owns points t, t + bs, ... and holds the first of its own maxima
    (sampling_gpu.cu:143-144); the shared-memory tree (:150-209) folds slot t + s into slot t for s = bs/2 ... 1 and keeps slot t
    on a tie, so the LAST fold decides between even and odd threads, the one before between t = 0 and 2 mod 4, ...: the winner
    is the thread whose BIT-REVERSED id is smallest (not the lowest id: threads 1 and 256 tie -> 256 wins), then the lowest point."""
    n = xyz.shape[0]
    bs = max(min(1 << int(np.floor(np.log2(n))), 1024), 1)
    bits = int(np.log2(bs))
    rev = np.array([int(format(t, f"0{bits}b")[::-1], 2) if bits else 0 for t in range(bs)], np.int64)
    temp = np.full(n, 1e10, np.float32)
    picks = np.zeros(m, np.int32)
    key = rev[np.arange(n) % bs] * n + np.arange(n)
    old = 0
    for j in range(1, m):
        temp = np.minimum(temp, d2_f32(xyz[old:old + 1], xyz)[0])
        cand = np.flatnonzero(temp == temp.max())
        old = int(cand[np.argmin(key[cand])])
        picks[j] = old
    return picks


@pytest.mark.parametrize("kind,n,m", [("cloud", 3000, 600), ("lattice", 2500, 2500), ("cloud", 700, 700)])
def test_fps_winner_and_tie_rule(kind, n, m):
    xyz = cloud(8, n) if kind == "cloud" else lattice(9, 12000)[:n]
    m = min(m, xyz.shape[0])
    got = oracle.furthest_point_sample(xyz[None], m)[0][0]
    assert np.array_equal(got, fps_by_rule(xyz, m))
    if m == xyz.shape[0]:
        assert sorted(got.tolist()) == list(range(m))          # sampling every point: a permutation


def test_roipoint_pool_takes_the_first_points_inside_in_index_order():
    from boxes_util import random_boxes
    pts, gt = synth.lidar_cloud(10, 4096, return_boxes=True)
    xyz = pts[None, :, :3].copy()
    feats = np.random.default_rng(0).standard_normal((1, 4096, 5)).astype(np.float32)
    boxes = np.concatenate([gt[:6, :7], random_boxes(np.random.default_rng(1), 3)[:, :7]]).astype(np.float32)[None]
    boxes[0, :, 3:6] += 1.0                                    # the caller enlarges them (POOL_EXTRA_WIDTH)
    ns = 64
    pooled, flag = oracle.roipoint_pool3d(xyz, feats, boxes, ns)
    inside = oracle.points_in_boxes_cpu(xyz[0], boxes[0])     # [boxes, points]: the CPU margin is wider, used only to bound the GPU-margin set
    for k in range(boxes.shape[1]):
        # independent statement of "inside": rotate into the box frame in float64, GPU margin 1e-5 (roipoint_pool3d_kernel.cu:16-36)
        b = boxes[0, k].astype(np.float64)
        rel = xyz[0].astype(np.float64) - b[:3]
        c, s = np.cos(-b[6]), np.sin(-b[6])
        lx, ly = rel[:, 0] * c - rel[:, 1] * s, rel[:, 0] * s + rel[:, 1] * c
        far = (np.abs(np.abs(rel[:, 2]) - b[5] / 2) > 1e-4) & (np.abs(np.abs(lx) - b[3] / 2) > 1e-4) & (np.abs(np.abs(ly) - b[4] / 2) > 1e-4)
        ins = (np.abs(rel[:, 2]) <= b[5] / 2) & (np.abs(lx) < b[3] / 2 + 1e-5) & (np.abs(ly) < b[4] / 2 + 1e-5)
        if not far.all():
            continue                                           # a point within 0.1 mm of a face: float32 vs float64 may disagree, skip the box
        idx = np.flatnonzero(ins)
        assert set(idx.tolist()) <= set(np.flatnonzero(inside[k]).tolist())
        if idx.size == 0:
            assert flag[0, k] == 1 and not pooled[0, k].any()
            continue
        take = np.resize(idx, ns) if idx.size < ns else idx[:ns]      # fewer than ns: the list repeats from its start
        assert flag[0, k] == 0
        assert np.array_equal(pooled[0, k, :, :3], xyz[0][take]) and np.array_equal(pooled[0, k, :, 3:], feats[0][take])


def test_interpolation_and_grouping_are_plain_indexing():
    rng = np.random.default_rng(3)
    f = rng.standard_normal((2, 6, 40)).astype(np.float32)
    idx = rng.integers(0, 40, (2, 25, 3)).astype(np.int32)
    w = rng.uniform(0, 1, (2, 25, 3)).astype(np.float32)
    out = oracle.three_interpolate_batch(f, idx, w)
    ref = np.einsum("bcnk,bnk->bcn", np.stack([f[b][:, idx[b]] for b in range(2)]).astype(np.float64), w.astype(np.float64))
    assert np.abs(out - ref).max() < 1e-5
    g = oracle.group_points_batch(f, idx)
    assert np.array_equal(g[1, 3, 7], f[1, 3, idx[1, 7]])
