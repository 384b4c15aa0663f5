"""Shared helpers for the spconv tests (random active sets, dense equivalents)."""
import numpy as np


def random_active(seed, batch, shape, n, sort=False):
    rng = np.random.default_rng(seed)
    vol = int(np.prod(shape))
    n = min(n, batch * vol)
    flat = rng.choice(batch * vol, size=n, replace=False)
    if sort:
        flat = np.sort(flat)
    b, r = flat // vol, flat % vol
    z, r = r // (shape[1] * shape[2]), r % (shape[1] * shape[2])
    y, x = r // shape[2], r % shape[2]
    return np.stack([b, z, y, x], 1).astype(np.int32)


def canon_pairs(pairs, num):
    """Sort each offset's pairs by input row (the canonical order of SURVEY §8 A3)."""
    out = []
    for k in range(pairs.shape[0]):
        p = pairs[k, :, : int(num[k])]
        order = np.argsort(p[0], kind="stable")
        out.append(p[:, order])
    return out


def voxel_indices_from_clouds(seeds, n_points=16384):
    """(b,z,y,x) int32 voxel coords of KITTI-shaped synthetic clouds, through the oracle voxeliser."""
    import oracle
    from fv2p_harness import synth
    inds = []
    for b, s in enumerate(seeds):
        pts = synth.lidar_cloud(s, n_points)
        _, c, _ = oracle.points_to_voxel(pts, synth.KITTI_VOXEL, synth.KITTI_RANGE, 5, 16000)
        inds.append(np.concatenate([np.full((c.shape[0], 1), b, np.int32), c], 1))
    return np.concatenate(inds, 0)
