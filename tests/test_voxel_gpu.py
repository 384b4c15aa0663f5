"""GPU parity of the hashed HIP voxeliser (A1) through the pcdet API and the C ABI.

Bar: every output is bit-exact — coords, per-voxel counts and the padded float payload (pure copies)."""
import glob
import os

import numpy as np
import pytest
import torch

import oracle
from fv2p_harness import synth
from pcdet.datasets.processor.voxel_generator import VoxelGenerator, points_to_voxel

pytestmark = pytest.mark.gpu
GOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "voxel_*.npz")))


def hip(fn, pts, *args):
    """The HIP entry point: points as a CUDA tensor (numpy arrays go to the library's host entry point, tests/test_voxel_host.py),
    results back as numpy."""
    out = fn(torch.from_numpy(np.ascontiguousarray(pts, dtype=np.float32)).cuda(), *args)
    assert all(t.is_cuda for t in out)
    return tuple(t.cpu().numpy() for t in out)


@pytest.mark.parametrize("path", GOLD, ids=lambda p: os.path.basename(p)[:-4])
def test_voxeliser_matches_reference_golden(gpu, path):
    g = np.load(path)
    gen = VoxelGenerator(g["voxel_size"], g["pc_range"], int(g["max_points"]), int(g["max_voxels"]))
    assert np.array_equal(gen.grid_size, g["grid_size"])
    v, c, n = hip(gen.generate, g["points"])
    assert c.dtype == np.int32 and n.dtype == np.int32 and v.dtype == np.float32
    assert np.array_equal(c, g["coors"])
    assert np.array_equal(n, g["num_points"])
    ov, _, _ = oracle.points_to_voxel(g["points"], g["voxel_size"], g["pc_range"], int(g["max_points"]), int(g["max_voxels"]))
    assert np.array_equal(v, ov)


@pytest.mark.parametrize("seed,n,mv,mp", [(1, 16384, 16000, 5), (2, 16384, 4000, 5), (3, 40000, 40000, 1), (4, 777, 20000, 35)])
def test_voxeliser_matches_oracle_on_seeded_clouds(gpu, seed, n, mv, mp):
    pts = synth.lidar_cloud(seed, n)
    v, c, k = hip(points_to_voxel, pts, synth.KITTI_VOXEL, synth.KITTI_RANGE, mp, True, mv)
    ov, oc, ok = oracle.points_to_voxel(pts, synth.KITTI_VOXEL, synth.KITTI_RANGE, mp, mv)
    assert np.array_equal(c, oc) and np.array_equal(k, ok) and np.array_equal(v, ov)


@pytest.mark.parametrize("seed", range(12))
def test_voxeliser_random_geometries_match_oracle(gpu, seed):
    """The HIP kernel on the random geometries of tests/test_voxel_host.py (random ranges, voxel sizes, feature widths, max_points /
    max_voxels with the break hit in about half of the cases, a fifth of the points exactly on cell faces, points outside the range):
    bit-exact against the oracle."""
    rng = np.random.default_rng(1000 + seed)
    lo = rng.uniform(-50, 0, 3).astype(np.float32)
    vs = rng.choice([0.05, 0.1, 0.16, 0.2, 0.4], 3).astype(np.float32)
    cells = rng.integers(3, 60, 3)
    rng_arr = np.concatenate([lo, lo + vs * cells]).astype(np.float32)
    n, ndim = int(rng.integers(1, 6000)), int(rng.integers(3, 7))
    pts = rng.uniform(-0.1, 1.1, (n, ndim)).astype(np.float32)
    pts[:, :3] = lo + pts[:, :3] * (rng_arr[3:] - lo)
    snap = rng.random(n) < 0.2
    pts[snap, :3] = (lo + np.round((pts[snap, :3] - lo) / vs) * vs).astype(np.float32)
    mp, mv = int(rng.integers(1, 9)), int(rng.integers(1, 400))
    v, c, k = hip(points_to_voxel, pts, vs, rng_arr, mp, True, mv)
    ov, oc, ok = oracle.points_to_voxel(pts, vs, rng_arr, mp, mv)
    assert np.array_equal(c, oc) and np.array_equal(k, ok) and np.array_equal(v, ov)


def test_voxeliser_waymo_shape_and_device_tensors(gpu):
    pts = synth.waymo_like_cloud(5, 180000)
    d = torch.from_numpy(pts).to(gpu)
    v, c, k = points_to_voxel(d, synth.WAYMO_VOXEL, synth.WAYMO_RANGE, 5, True, 80000)
    assert v.is_cuda and c.is_cuda and k.is_cuda
    ov, oc, ok = oracle.points_to_voxel(pts, synth.WAYMO_VOXEL, synth.WAYMO_RANGE, 5, 80000)
    assert np.array_equal(c.cpu().numpy(), oc) and np.array_equal(k.cpu().numpy(), ok)
    assert np.array_equal(v.cpu().numpy(), ov)
    # reverse_index=False only flips the coordinate columns (voxel_generator.py:119-127 vs :210-279)
    v2, c2, k2 = points_to_voxel(d, synth.WAYMO_VOXEL, synth.WAYMO_RANGE, 5, False, 80000)
    assert torch.equal(c2, c.flip(1)) and torch.equal(k2, k) and torch.equal(v2, v)


def test_voxeliser_edge_cases(gpu):
    # empty input
    v, c, k = hip(points_to_voxel, np.zeros((0, 4), np.float32), synth.KITTI_VOXEL, synth.KITTI_RANGE, 5, True, 100)
    assert v.shape == (0, 5, 4) and c.shape == (0, 3) and k.shape == (0,)
    # everything out of range
    pts = np.full((100, 4), 1000.0, np.float32)
    v, c, k = hip(points_to_voxel, pts, synth.KITTI_VOXEL, synth.KITTI_RANGE, 5, True, 100)
    assert v.shape[0] == 0
    # all points in ONE voxel, more than max_points
    pts = np.tile(np.array([[10.01, 0.01, -1.01, 0.5]], np.float32), (1000, 1))
    pts[:, 3] = np.arange(1000)
    v, c, k = hip(points_to_voxel, pts, synth.KITTI_VOXEL, synth.KITTI_RANGE, 5, True, 100)
    assert v.shape[0] == 1 and int(k[0]) == 5 and np.array_equal(v[0, :, 3], np.arange(5, dtype=np.float32))
    # max_voxels = 1: the second distinct voxel breaks the scan, later points of voxel 0 are lost too
    pts = np.array([[10.01, 0.01, -1.01, 0], [20.0, 0.0, -1.0, 1], [10.02, 0.01, -1.01, 2]], np.float32)
    v, c, k = hip(points_to_voxel, pts, synth.KITTI_VOXEL, synth.KITTI_RANGE, 5, True, 1)
    ov, oc, ok = oracle.points_to_voxel(pts, synth.KITTI_VOXEL, synth.KITTI_RANGE, 5, 1)
    assert np.array_equal(k, ok) and int(k[0]) == 1 and np.array_equal(v, ov)


def test_cpu_tensor_is_rejected_loudly(gpu):
    with pytest.raises(Exception):
        points_to_voxel(torch.zeros(10, 4), synth.KITTI_VOXEL, synth.KITTI_RANGE, 5, True, 100)


def test_batched_multistream_voxeliser_equals_per_cloud(gpu):
    from pcdet.datasets.processor.voxel_generator import points_to_voxel_batch
    clouds = [synth.lidar_cloud(20 + b, 16384 if b % 2 == 0 else 9000) for b in range(4)]
    v, c, k = points_to_voxel_batch([torch.from_numpy(p).to(gpu) for p in clouds], synth.KITTI_VOXEL, synth.KITTI_RANGE, 5, 16000)
    ref = [oracle.points_to_voxel(p, synth.KITTI_VOXEL, synth.KITTI_RANGE, 5, 16000) for p in clouds]
    assert np.array_equal(v.cpu().numpy(), np.concatenate([r[0] for r in ref]))
    assert np.array_equal(k.cpu().numpy(), np.concatenate([r[2] for r in ref]))
    rc = np.concatenate([np.concatenate([np.full((r[1].shape[0], 1), b, np.int32), r[1]], 1) for b, r in enumerate(ref)])
    assert c.dtype == torch.int32 and np.array_equal(c.cpu().numpy(), rc)


def test_mean_vfe_collate_matches_padded_batch_expression(gpu):
    """points_to_voxel_batch(mean_vfe=True) (fv2p_voxel_mean_collate) equals MeanVFE over the padded batch it replaces:
    features = sum over the point slots / clamp_min(num_points, 1) (vfe/mean_vfe.py:14-31), coords = (b, z, y, x)."""
    import torch
    from fv2p_harness import synth
    from pcdet.datasets.processor.voxel_generator import points_to_voxel_batch
    clouds = [torch.from_numpy(synth.lidar_cloud(7 + b, 6000 + 500 * b)).to(gpu) for b in range(3)]
    v, c, k = points_to_voxel_batch(clouds, synth.KITTI_VOXEL, synth.KITTI_RANGE, 5, 16000)
    want = v.sum(dim=1) / torch.clamp_min(k.view(-1, 1), 1.0).type_as(v)
    feats, coords = points_to_voxel_batch(clouds, synth.KITTI_VOXEL, synth.KITTI_RANGE, 5, 16000, mean_vfe=True)
    assert torch.equal(coords, c) and coords.dtype == torch.int32
    assert feats.shape == want.shape and torch.allclose(feats, want, rtol=1e-6, atol=0)
    # max_voxels overflow: only the kept voxels are collated
    f2, c2 = points_to_voxel_batch(clouds, synth.KITTI_VOXEL, synth.KITTI_RANGE, 5, 1000, mean_vfe=True)
    v2, cc2, k2 = points_to_voxel_batch(clouds, synth.KITTI_VOXEL, synth.KITTI_RANGE, 5, 1000)
    assert f2.shape[0] == v2.shape[0] == 3000 and torch.equal(c2, cc2)
