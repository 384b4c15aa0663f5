"""Seeded LiDAR-like synthetic clouds (SURVEY.md §8d "Synthetic inputs").

A uniform random cloud is useless here: it yields one point per voxel and trips the
`max_voxels` early break of the reference voxeliser (voxel_generator.py:198-199).  This
generator casts a 64-beam spinning-LiDAR ray fan against a ground plane, 20-40 car-sized
boxes and a few vertical walls, adds 2 cm range noise, crops to the detection range and
subsamples/shuffles to a fixed point count.
"""
import numpy as np

KITTI_RANGE = np.array([0.0, -40.0, -3.0, 70.4, 40.0, 1.0], dtype=np.float32)
KITTI_VOXEL = np.array([0.05, 0.05, 0.1], dtype=np.float32)
WAYMO_RANGE = np.array([-75.2, -75.2, -2.0, 75.2, 75.2, 4.0], dtype=np.float32)
WAYMO_VOXEL = np.array([0.1, 0.1, 0.15], dtype=np.float32)


def _ray_box(dirs, center, size, yaw):
    """Entry distance of unit rays from the origin into a yawed box (inf if missed)."""
    c, s = np.cos(-yaw), np.sin(-yaw)
    rot = np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])
    d = dirs @ rot.T
    o = -(rot @ center)
    half = size / 2.0
    with np.errstate(divide="ignore", invalid="ignore"):
        t1 = (-half - o) / d
        t2 = (half - o) / d
    tmin = np.nanmax(np.minimum(t1, t2), axis=1)
    tmax = np.nanmin(np.maximum(t1, t2), axis=1)
    hit = (tmax >= np.maximum(tmin, 0.0)) & (tmin > 0.5)
    return np.where(hit, tmin, np.inf)


def lidar_cloud(seed, n_points=16384, pc_range=KITTI_RANGE, fov_deg=45.0, n_beams=64,
                az_step_deg=0.09, sensor_ground=-1.73, return_boxes=False):
    """Returns points [n_points, 4] float32 (x, y, z, intensity); optionally gt boxes [G, 7]."""
    rng = np.random.default_rng(seed)
    elev = np.deg2rad(np.linspace(-24.8, 2.0, n_beams))
    az = np.deg2rad(np.arange(-fov_deg, fov_deg, az_step_deg))
    az = az + rng.uniform(0, np.deg2rad(az_step_deg))
    ee, aa = np.meshgrid(elev, az, indexing="ij")
    dirs = np.stack([np.cos(ee) * np.cos(aa), np.cos(ee) * np.sin(aa), np.sin(ee)], axis=-1).reshape(-1, 3)
    with np.errstate(divide="ignore"):
        t = np.where(dirs[:, 2] < -1e-3, sensor_ground / dirs[:, 2], np.inf)
    # cars
    n_box = int(rng.integers(20, 41))
    xr = (max(pc_range[0], -70.0) + 4.0, pc_range[3] - 4.0)
    yr = (pc_range[1] + 3.0, pc_range[4] - 3.0)
    boxes = []
    for _ in range(n_box):
        cx, cy = rng.uniform(*xr), rng.uniform(*yr)
        if abs(cx) < 3.0 and abs(cy) < 3.0:
            cx += 6.0
        size = np.array([3.9, 1.6, 1.56]) * rng.uniform(0.9, 1.15, size=3)
        yaw = rng.uniform(-np.pi, np.pi)
        center = np.array([cx, cy, sensor_ground + size[2] / 2.0])
        boxes.append(np.concatenate([center, size, [yaw]]))
        t = np.minimum(t, _ray_box(dirs, center, size, yaw))
    # walls: tall thin boxes
    for _ in range(int(rng.integers(3, 7))):
        cx, cy = rng.uniform(*xr), rng.uniform(*yr)
        size = np.array([rng.uniform(8, 30), 0.3, rng.uniform(2.0, 4.5)])
        yaw = rng.uniform(-np.pi, np.pi)
        center = np.array([cx, cy, sensor_ground + size[2] / 2.0])
        if np.hypot(cx, cy) > 8.0:
            t = np.minimum(t, _ray_box(dirs, center, size, yaw))
    ok = np.isfinite(t) & (t < 120.0)
    t = t[ok] + rng.normal(0.0, 0.02, size=int(ok.sum()))
    pts = dirs[ok] * t[:, None]
    lo, hi = pc_range[:3], pc_range[3:]
    m = np.all((pts >= lo) & (pts <= hi), axis=1)
    pts = pts[m]
    inten = rng.uniform(0.0, 1.0, size=(pts.shape[0], 1))
    pts = np.concatenate([pts, inten], axis=1).astype(np.float32)
    perm = rng.permutation(pts.shape[0])
    if pts.shape[0] >= n_points:
        pts = pts[perm[:n_points]]
    else:  # pad by re-sampling with jitter so voxels stay multi-point
        extra = pts[rng.integers(0, pts.shape[0], size=n_points - pts.shape[0])].copy()
        extra[:, :3] += rng.normal(0, 0.01, size=(extra.shape[0], 3)).astype(np.float32)
        pts = np.concatenate([pts[perm], extra], axis=0)
        pts = pts[rng.permutation(n_points)]
    pts = np.ascontiguousarray(pts, dtype=np.float32)
    if return_boxes:
        return pts, np.asarray(boxes, dtype=np.float32)
    return pts


def waymo_like_cloud(seed, n_points=180000, return_boxes=False):
    return lidar_cloud(seed, n_points=n_points, pc_range=WAYMO_RANGE, fov_deg=180.0, az_step_deg=0.13,
                       return_boxes=return_boxes)


def proposal_boxes(seed, n, spread=35.0, tight=False):
    """n rotated car-sized boxes clustered around n/12 centres, like a first-stage detector's output before NMS.
    tight=True: what a trained first stage emits — n/60 objects, each proposed ~60 times with 15 cm / 3 degrees of jitter, nearly equal
    sizes, so that most boxes overlap an earlier one above the NMS threshold (~1200 of 9000 survive at 0.8)."""
    rng = np.random.default_rng(seed)
    if tight:
        k = max(n // 60, 1)
        centers = rng.uniform(-spread, spread, size=(k, 2))
        which = rng.integers(0, k, n)
        xy = centers[which] + rng.normal(0, 0.15, size=(n, 2))
        z = rng.uniform(-1.2, -0.8, size=(k, 1))[which] + rng.normal(0, 0.02, size=(n, 1))
        dims = (np.array([3.9, 1.6, 1.56]) * rng.uniform(0.9, 1.1, size=(k, 3)))[which] * rng.uniform(0.98, 1.02, size=(n, 3))
        yaw = rng.uniform(-np.pi, np.pi, size=(k, 1))[which] + rng.normal(0, 0.05, size=(n, 1))
        return np.concatenate([xy, z, dims, yaw], 1).astype(np.float32)
    centers = rng.uniform(-spread, spread, size=(max(n // 12, 1), 2))
    xy = centers[rng.integers(0, centers.shape[0], n)] + rng.normal(0, 0.6, size=(n, 2))
    z = rng.uniform(-1.5, 0.5, size=(n, 1))
    dims = np.array([3.9, 1.6, 1.56]) * rng.uniform(0.7, 1.3, size=(n, 3))
    yaw = rng.uniform(-np.pi, np.pi, size=(n, 1))
    return np.concatenate([xy, z, dims, yaw], 1).astype(np.float32)
