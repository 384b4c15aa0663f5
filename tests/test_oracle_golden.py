"""CPU: the oracle restatements reproduce the golden vectors generated from the reference itself."""
import glob
import hashlib
import os

import numpy as np
import pytest

import oracle


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "voxel_*.npz"))),
                         ids=lambda p: os.path.basename(p)[:-4])
def test_voxel_oracle_matches_reference_golden(path):
    g = np.load(path)
    v, c, n = oracle.points_to_voxel(g["points"], g["voxel_size"], g["pc_range"], int(g["max_points"]), int(g["max_voxels"]))
    assert np.array_equal(c, g["coors"])
    assert np.array_equal(n, g["num_points"])
    assert hashlib.sha256(np.ascontiguousarray(v).tobytes()).digest() == g["voxels_sha256"].tobytes()
    if "voxels" in g.files:
        assert np.array_equal(v, g["voxels"])


def test_voxel_oracle_scratch_map_equals_dense_map():
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "voxel_kitti2k_break.npz"))
    grid = g["grid_size"]
    scratch = -np.ones(int(np.prod(grid)), np.int32)
    a = oracle.points_to_voxel(g["points"], g["voxel_size"], g["pc_range"], 5, 300)
    b = oracle.points_to_voxel(g["points"], g["voxel_size"], g["pc_range"], 5, 300, scratch_map=scratch)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    assert (scratch == -1).all()


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "bev_interp_*.npz"))),
                         ids=lambda p: os.path.basename(p)[:-4])
def test_bev_interp_oracle_matches_reference_golden(path):
    """oracle/bev_oracle.py against the outputs of the reference's own bilinear_interpolate_torch (fixtures made by
    oracle/gen_golden_bev.py in the build container): bit-exact, including points on and beyond the map border."""
    from oracle import bev_oracle
    d = np.load(path)
    assert np.array_equal(bev_oracle.bilinear_interpolate(d["im"], d["x"], d["y"]), d["out"])
