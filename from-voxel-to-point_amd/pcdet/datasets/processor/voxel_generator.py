"""Voxeliser behind the reference's `VoxelGenerator` / `points_to_voxel` API.

Mirrors pcdet/datasets/processor/voxel_generator.py:5-133 of the reference (class
`VoxelGenerator`, function `points_to_voxel`): same constructor arguments, properties,
return triple `(voxels [M,max_points,ndim] f32, coordinates [M,3] i32 (z,y,x), num_points [M] i32)`
and the same first-come / max_voxels-break semantics.  Two entry points of the native library, chosen by
WHERE THE DATA IS, never by what happens to be available:

* CUDA tensor in -> CUDA tensors out: the hashed HIP voxeliser `fv2p_points_to_voxel` (csrc/voxelize.hip); this is the path of the
  models and of bench.py (`points_to_voxel_batch`).
* numpy in -> numpy out: `fv2p_points_to_voxel_host`, the reference's own call site - `VoxelGenerator.generate` runs on numpy arrays in
  forked DataLoader worker processes (data_processor.py:43-81), where HIP cannot be initialised.  Like `points_in_boxes_cpu` and
  `boxes_bev_iou_cpu` it is the reference's CPU entry point served by the library (host pointers, the calling thread, no HIP call;
  importing this module does not initialise HIP).  Same results bit for bit (tests/test_oracle_golden.py holds both to the reference's
  fixtures).  There is no fallback in either direction: a CUDA tensor never goes to the host function, and without the library both raise.
"""
import numpy as np
import torch

from pcdet import ops as _ops  # noqa: F401  (puts fv2p_native on sys.path)
import fv2p_native as _nat


def _grid_size(voxel_size, coors_range):
    # float32 arithmetic + np.round, as voxel_generator.py:22-27,104-105
    g = (coors_range[3:] - coors_range[:3]) / voxel_size
    return np.round(g).astype(np.int64)


def points_to_voxel_gpu(points, voxel_size, coors_range, max_points=35, reverse_index=True, max_voxels=20000):
    """points: CUDA float32 tensor [N, ndim>=3]. Returns CUDA tensors sliced to the voxel count."""
    import torch
    _nat.require_cuda(points)
    if points.dtype != torch.float32:
        raise TypeError("points must be float32")
    points = points.contiguous()
    voxel_size = np.asarray(voxel_size, dtype=np.float32)
    coors_range = np.asarray(coors_range, dtype=np.float32)
    grid = _grid_size(voxel_size, coors_range)
    n, ndim = points.shape
    dev = points.device
    voxels = torch.empty((max_voxels, max_points, ndim), dtype=torch.float32, device=dev)
    coors = torch.empty((max_voxels, 3), dtype=torch.int32, device=dev)
    num = torch.empty((max_voxels,), dtype=torch.int32, device=dev)
    count = torch.empty((1,), dtype=torch.int32, device=dev)
    with _nat.device_guard(dev):
        ws_bytes = _nat.lib().fv2p_points_to_voxel_ws_bytes(n, max_voxels)
        ws = _nat.workspace(ws_bytes, dev)
        _nat.call("fv2p_points_to_voxel", points, n, ndim, voxel_size.tolist(), coors_range[:3].tolist(),
                  [int(g) for g in grid], int(max_points), int(max_voxels), voxels, coors, num, count,
                  ws, ws.numel(), _nat.stream())
    m = int(count.item())
    voxels, coors, num = voxels[:m], coors[:m], num[:m]
    if not reverse_index:
        coors = coors.flip(1).contiguous()
    return voxels, coors, num


_STREAMS = {}


def _launch(points, voxel_size, coors_range, grid, max_points, max_voxels):
    """Enqueues one cloud's voxelisation on the current stream; no host synchronisation."""
    n, ndim = points.shape
    dev = points.device
    voxels = torch.empty((max_voxels, max_points, ndim), dtype=torch.float32, device=dev)
    coors = torch.empty((max_voxels, 3), dtype=torch.int32, device=dev)
    num = torch.empty((max_voxels,), dtype=torch.int32, device=dev)
    count = torch.empty((1,), dtype=torch.int32, device=dev)
    ws_bytes = _nat.lib().fv2p_points_to_voxel_ws_bytes(n, max_voxels)
    ws = _nat.workspace(ws_bytes, dev)
    _nat.call("fv2p_points_to_voxel", points, n, ndim, voxel_size.tolist(), coors_range[:3].tolist(), [int(g) for g in grid],
              int(max_points), int(max_voxels), voxels, coors, num, count, ws, ws.numel(), _nat.stream())
    return voxels, coors, num, count


def points_to_voxel_batch(points_list, voxel_size, coors_range, max_points=35, max_voxels=20000, mean_vfe=False, cloud_streams=True):
    """Voxelises the clouds of one batch concurrently — one HIP stream per cloud, a single host sync for all voxel
    counts — and returns the collated batch the models consume (dataset.collate_batch, pcdet/datasets/dataset.py:152-183):
    voxels [sum M, max_points, ndim], coords [sum M, 4] (batch, z, y, x) int32, num_points [sum M] int32.
    mean_vfe=True returns (features [sum M, ndim], coords) instead: MeanVFE (vfe/mean_vfe.py:14-31) and the collation
    in one launch per cloud, without the padded [M, max_points, ndim] batch tensor."""
    voxel_size = np.asarray(voxel_size, dtype=np.float32)
    coors_range = np.asarray(coors_range, dtype=np.float32)
    grid = _grid_size(voxel_size, coors_range)
    dev = points_list[0].device
    ext = _nat.torch_ext() if mean_vfe else None
    if ext is not None and all(p.is_cuda and p.dtype == torch.float32 and p.dim() == 2 for p in points_list):
        # the whole batch (per-cloud streams, the one synchronisation, MeanVFE + collate) inside the compiled binding,
        # without the interpreter lock: an input-pipeline thread then barely competes with the training thread for it
        with _nat.device_guard(dev):
            feats, coords = ext.voxelize_batch_mean(list(points_list), [float(v) for v in voxel_size], [float(v) for v in coors_range[:3]],
                                                    [int(g) for g in grid], int(max_points), int(max_voxels), bool(cloud_streams))
        return feats, coords
    main = torch.cuda.current_stream(dev)
    pool = _STREAMS.setdefault((dev.index, main.cuda_stream), [])   # per calling stream: concurrent pipeline threads never share
    while len(pool) < len(points_list):
        pool.append(torch.cuda.Stream(device=dev))
    outs = []
    with _nat.device_guard(dev):
        for pts, st in zip(points_list, pool):
            _nat.require_cuda(pts)
            st.wait_stream(main)
            with torch.cuda.stream(st):
                outs.append(_launch(pts.contiguous(), voxel_size, coors_range, grid, max_points, max_voxels))
        for st, o in zip(pool, outs):
            main.wait_stream(st)
            for t in o:
                t.record_stream(main)
    counts = torch.cat([o[3] for o in outs]).cpu().tolist()  # the one synchronisation of the batch
    if mean_vfe:
        counts = [min(int(m), int(max_voxels)) for m in counts]
        total, ndim = sum(counts), points_list[0].shape[1]
        feats = torch.empty((total, ndim), dtype=torch.float32, device=dev)
        coords = torch.empty((total, 4), dtype=torch.int32, device=dev)
        off = 0
        with _nat.device_guard(dev):
            for b, (o, m) in enumerate(zip(outs, counts)):
                if m > 0:
                    _nat.call("fv2p_voxel_mean_collate", o[0], o[1], o[2], o[3], m, int(max_points), ndim, b,
                              feats[off:], coords[off:], _nat.stream())
                off += m
        return feats, coords
    v = torch.cat([o[0][:m] for o, m in zip(outs, counts)])
    c = torch.cat([torch.nn.functional.pad(o[1][:m], (1, 0), value=b) for b, (o, m) in enumerate(zip(outs, counts))])
    k = torch.cat([o[2][:m] for o, m in zip(outs, counts)])
    return v, c, k


def points_to_voxel(points, voxel_size, coors_range, max_points=35, reverse_index=True, max_voxels=20000):
    """Drop-in for the reference function of the same name (voxel_generator.py:75-133)."""
    import torch
    if isinstance(points, torch.Tensor):
        return points_to_voxel_gpu(points, voxel_size, coors_range, max_points, reverse_index, max_voxels)
    pts = np.ascontiguousarray(points)
    if not isinstance(voxel_size, np.ndarray):
        voxel_size = np.array(voxel_size, dtype=pts.dtype)
    if not isinstance(coors_range, np.ndarray):
        coors_range = np.array(coors_range, dtype=pts.dtype)
    v, c, n = points_to_voxel_host(pts.astype(np.float32, copy=False), voxel_size, coors_range, max_points, reverse_index, max_voxels)
    return v.astype(pts.dtype, copy=False), c, n


def points_to_voxel_host(points, voxel_size, coors_range, max_points=35, reverse_index=True, max_voxels=20000):
    """numpy float32 [N, ndim >= 3] -> numpy (voxels, coors, num_points): the library's host entry point (no HIP call)."""
    pts = np.ascontiguousarray(points, dtype=np.float32)
    if pts.ndim != 2 or pts.shape[1] < 3:
        raise ValueError("points must be [N, ndim >= 3]")
    voxel_size = np.asarray(voxel_size, dtype=np.float32)
    coors_range = np.asarray(coors_range, dtype=np.float32)
    grid = _grid_size(voxel_size, coors_range)
    n, ndim = pts.shape
    voxels = torch.empty((max_voxels, max_points, ndim), dtype=torch.float32)
    coors = torch.empty((max_voxels, 3), dtype=torch.int32)
    num = torch.empty((max_voxels,), dtype=torch.int32)
    count = torch.zeros((1,), dtype=torch.int32)
    _nat.call("fv2p_points_to_voxel_host", torch.from_numpy(pts), n, ndim, voxel_size.tolist(), coors_range[:3].tolist(),
              [int(g) for g in grid], int(max_points), int(max_voxels), voxels, coors, num, count)
    m = int(count[0])
    v, c, k = voxels[:m].numpy(), coors[:m].numpy(), num[:m].numpy()
    if not reverse_index:
        c = np.ascontiguousarray(c[:, ::-1])
    return v, c, k


class VoxelGenerator(object):
    """Same surface as the reference class (voxel_generator.py:5-72)."""

    def __init__(self, voxel_size, point_cloud_range, max_num_points, max_voxels=20000):
        point_cloud_range = np.array(point_cloud_range, dtype=np.float32)
        voxel_size = np.array(voxel_size, dtype=np.float32)
        self._voxel_size = voxel_size
        self._point_cloud_range = point_cloud_range
        self._max_num_points = max_num_points
        self._max_voxels = max_voxels
        self._grid_size = _grid_size(voxel_size, point_cloud_range)

    def generate(self, points):
        return points_to_voxel(points, self._voxel_size, self._point_cloud_range, self._max_num_points, True,
                               self._max_voxels)

    @property
    def voxel_size(self):
        return self._voxel_size

    @property
    def max_num_points_per_voxel(self):
        return self._max_num_points

    @property
    def point_cloud_range(self):
        return self._point_cloud_range

    @property
    def grid_size(self):
        return self._grid_size

    def __repr__(self):
        return (f"{self.__class__.__name__}(voxel_size={self._voxel_size}, "
                f"point_cloud_range={self._point_cloud_range.tolist()}, max_num_points={self._max_num_points}, "
                f"max_voxels={self._max_voxels}, grid_size={self._grid_size.tolist()})")
