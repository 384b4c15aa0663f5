"""A1 at the reference's own call site (no GPU): `VoxelGenerator.generate` on numpy arrays, as the reference's DataProcessor calls it
inside forked DataLoader workers (pcdet/datasets/processor/data_processor.py:43-81 -> voxel_generator.py:75-207).  The library serves it
on the host (`fv2p_points_to_voxel_host`: host pointers, the calling thread, no HIP call) - the counterpart of BASELINE configs[0]
("points_to_voxel ... CPU reference path, index bit-exactness, no GPU").  Held to the six fixtures written by the reference's own
function (tests/golden/voxel_*.npz, oracle/gen_golden_voxel.py) and to the oracle; CUDA tensors take the HIP kernel instead
(tests/test_voxel_gpu.py) and CPU torch tensors are refused."""
import glob
import hashlib
import os

import numpy as np
import pytest
import torch

import oracle
from fv2p_harness import synth
from pcdet.datasets.processor.voxel_generator import VoxelGenerator, points_to_voxel

GOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "voxel_*.npz")))


@pytest.mark.parametrize("path", GOLD, ids=lambda p: os.path.basename(p)[:-4])
def test_host_voxeliser_matches_reference_golden(path):
    g = np.load(path)
    gen = VoxelGenerator(g["voxel_size"], g["pc_range"], int(g["max_points"]), int(g["max_voxels"]))
    assert np.array_equal(gen.grid_size, g["grid_size"])
    v, c, n = gen.generate(g["points"])
    assert isinstance(v, np.ndarray) and c.dtype == np.int32 and n.dtype == np.int32 and v.dtype == np.float32
    assert np.array_equal(c, g["coors"]) and np.array_equal(n, g["num_points"])
    assert hashlib.sha256(np.ascontiguousarray(v).tobytes()).digest() == g["voxels_sha256"].tobytes()
    if "voxels" in g.files:
        assert np.array_equal(v, g["voxels"])


@pytest.mark.parametrize("seed,n,mv,mp", [(1, 16384, 16000, 5), (2, 16384, 4000, 5), (3, 40000, 40000, 1), (4, 777, 20000, 35)])
def test_host_voxeliser_matches_oracle_on_seeded_clouds(seed, n, mv, mp):
    pts = synth.lidar_cloud(seed, n)
    v, c, k = points_to_voxel(pts, synth.KITTI_VOXEL, synth.KITTI_RANGE, mp, True, mv)
    ov, oc, ok = oracle.points_to_voxel(pts, synth.KITTI_VOXEL, synth.KITTI_RANGE, mp, mv)
    assert np.array_equal(c, oc) and np.array_equal(k, ok) and np.array_equal(v, ov)
    v2, c2, k2 = points_to_voxel(pts, synth.KITTI_VOXEL, synth.KITTI_RANGE, mp, False, mv)     # voxel_generator.py:119-127: (x, y, z) columns
    assert np.array_equal(c2, c[:, ::-1]) and np.array_equal(k2, k) and np.array_equal(v2, v)


@pytest.mark.parametrize("seed", range(12))
def test_host_voxeliser_random_geometries_match_oracle(seed):
    """Random ranges, voxel sizes, point counts, feature widths, max_points / max_voxels (the max_voxels break hit in about half of the
    cases), points on cell faces and outside the range: coordinates, counts and payload bit-exact against the oracle's restatement of
    voxel_generator.py:136-207."""
    rng = np.random.default_rng(1000 + seed)
    lo = rng.uniform(-50, 0, 3).astype(np.float32)
    vs = rng.choice([0.05, 0.1, 0.16, 0.2, 0.4], 3).astype(np.float32)
    cells = rng.integers(3, 60, 3)
    rng_arr = np.concatenate([lo, lo + vs * cells]).astype(np.float32)
    n, ndim = int(rng.integers(1, 6000)), int(rng.integers(3, 7))
    pts = rng.uniform(-0.1, 1.1, (n, ndim)).astype(np.float32)
    pts[:, :3] = lo + pts[:, :3] * (rng_arr[3:] - lo)
    snap = rng.random(n) < 0.2                                       # a fifth of the points exactly on cell faces
    pts[snap, :3] = (lo + np.round((pts[snap, :3] - lo) / vs) * vs).astype(np.float32)
    mp, mv = int(rng.integers(1, 9)), int(rng.integers(1, 400))
    v, c, k = points_to_voxel(pts, vs, rng_arr, mp, True, mv)
    ov, oc, ok = oracle.points_to_voxel(pts, vs, rng_arr, mp, mv)
    assert np.array_equal(c, oc) and np.array_equal(k, ok) and np.array_equal(v, ov)
    assert v.shape[0] <= mv and (k >= 1).all() and (k <= mp).all()


def test_host_voxeliser_edge_cases_and_dtypes():
    v, c, k = points_to_voxel(np.zeros((0, 4), np.float32), synth.KITTI_VOXEL, synth.KITTI_RANGE, 5, True, 100)
    assert v.shape == (0, 5, 4) and c.shape == (0, 3) and k.shape == (0,)
    v, c, k = points_to_voxel(np.full((100, 4), 1000.0, np.float32), synth.KITTI_VOXEL, synth.KITTI_RANGE, 5, True, 100)
    assert v.shape[0] == 0
    pts = np.tile(np.array([[10.01, 0.01, -1.01, 0.5]], np.float32), (1000, 1))
    pts[:, 3] = np.arange(1000)
    v, c, k = points_to_voxel(pts, synth.KITTI_VOXEL, synth.KITTI_RANGE, 5, True, 100)
    assert v.shape[0] == 1 and int(k[0]) == 5 and np.array_equal(v[0, :, 3], np.arange(5, dtype=np.float32))
    # max_voxels = 1: the second distinct voxel stops the scan, later points of voxel 0 are lost too (voxel_generator.py:198-199)
    pts = np.array([[10.01, 0.01, -1.01, 0], [20.0, 0.0, -1.0, 1], [10.02, 0.01, -1.01, 2]], np.float32)
    v, c, k = points_to_voxel(pts, synth.KITTI_VOXEL, synth.KITTI_RANGE, 5, True, 1)
    assert int(k[0]) == 1 and v.shape[0] == 1
    # the upper range bound itself is outside the grid (Appendix A.1), float64 input comes back as float64 (the reference keeps the dtype)
    edge = np.array([[70.4, 0.0, 0.0, 1.0], [70.39, 39.99, 0.99, 2.0]], np.float64)
    v, c, k = points_to_voxel(edge, synth.KITTI_VOXEL, synth.KITTI_RANGE, 5, True, 10)
    assert v.dtype == np.float64 and v.shape[0] == 1 and c[0].tolist() == [39, 1599, 1407]
    with pytest.raises(Exception):
        points_to_voxel(torch.zeros(10, 4), synth.KITTI_VOXEL, synth.KITTI_RANGE, 5, True, 100)    # CPU torch tensor: refused, not rerouted


_WORKER_SCRIPT = r"""
import sys
import numpy as np
import torch
import oracle
from fv2p_harness import synth
from pcdet.datasets.processor.voxel_generator import VoxelGenerator


class Clouds(torch.utils.data.Dataset):
    # what the reference's dataset does per sample: voxelise inside __getitem__ (DataProcessor.transform_points_to_voxels)
    def __init__(self):
        self.gen = VoxelGenerator(synth.KITTI_VOXEL, synth.KITTI_RANGE, 5, 16000)

    def __len__(self):
        return 4

    def __getitem__(self, i):
        v, c, k = self.gen.generate(synth.lidar_cloud(50 + i, 4096))
        return {"voxels": v, "coors": c, "num": k, "hip": torch.cuda.is_initialized()}


assert not torch.cuda.is_initialized(), "importing pcdet.ops / the voxel generator must not initialise HIP"
loader = torch.utils.data.DataLoader(Clouds(), batch_size=1, num_workers=2, multiprocessing_context="fork", collate_fn=lambda b: b[0])
seen = 0
for i, item in enumerate(loader):
    ov, oc, ok = oracle.points_to_voxel(synth.lidar_cloud(50 + i, 4096), synth.KITTI_VOXEL, synth.KITTI_RANGE, 5, 16000)
    assert np.array_equal(item["coors"], oc) and np.array_equal(item["num"], ok) and np.array_equal(item["voxels"], ov)
    assert item["hip"] is False
    seen += 1
assert seen == 4 and not torch.cuda.is_initialized()
print("WORKERS OK")
"""


def test_generate_runs_in_forked_dataloader_workers_without_hip():
    """In a process of its own (whatever this pytest process has initialised by now): importing the package does not initialise HIP, two
    FORKED DataLoader workers voxelise numpy clouds through `VoxelGenerator.generate` bit-exactly, and nobody has initialised HIP at the end."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([root, os.path.join(root, "from-voxel-to-point_amd"), os.environ.get("PYTHONPATH", "")]))
    out = subprocess.run([sys.executable, "-c", _WORKER_SCRIPT], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0 and "WORKERS OK" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
