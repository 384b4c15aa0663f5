"""points_in_boxes_{cpu,gpu} and RoIAwarePool3d behind the names of the reference's
pcdet/ops/roiaware_pool3d/roiaware_pool3d_utils.py:9-107, on the C ABI (pcdet/ops/_glue.py)."""
import numpy as np
import torch
import torch.nn as nn

import fv2p_native as _nat

from .. import _glue as G

_POOL_CODE = {"max": 0, "avg": 1}


def points_in_boxes_cpu(points, boxes):
    """points (M, 3), boxes (N, 7) [x, y, z, dx, dy, dz, heading] -> (N, M) int32 0/1 matrix, evaluated on the host with the
    reference's 1e-2 margin; numpy in -> numpy out."""
    as_numpy = isinstance(points, np.ndarray) or isinstance(boxes, np.ndarray)
    pts = torch.as_tensor(points).float().contiguous()
    bxs = torch.as_tensor(boxes).float().contiguous()
    if bxs.shape[1] != 7 or pts.shape[1] != 3:
        raise AssertionError("points_in_boxes_cpu: points (M, 3), boxes (N, 7)")
    if pts.is_cuda or bxs.is_cuda:
        raise _nat.Fv2pError("points_in_boxes_cpu takes CPU tensors")
    flags = torch.zeros((bxs.shape[0], pts.shape[0]), dtype=torch.int32)
    _nat.call("fv2p_points_in_boxes_cpu", bxs, pts, bxs.shape[0], pts.shape[0], flags)
    return flags.numpy() if as_numpy else flags


def points_in_boxes_gpu(points, boxes):
    """points (B, M, 3), boxes (B, T, 7) -> (B, M) int32: index of the first box containing the point, -1 for background."""
    if boxes.shape[0] != points.shape[0] or boxes.shape[2] != 7 or points.shape[2] != 3:
        raise AssertionError("points_in_boxes_gpu: points (B, M, 3), boxes (B, T, 7)")
    b, m, _ = points.shape
    owner = torch.full((b, m), -1, dtype=torch.int32, device=points.device)
    G.run("fv2p_points_in_boxes", boxes.contiguous(), points.contiguous(), b, boxes.shape[1], m, owner)
    return owner


def _pool(saved, rois, pts, pts_feature, out_size, max_pts_each_voxel, pool_method):
    """rois (N, 7), pts (P, 3), pts_feature (P, C) -> (N, ox, oy, oz, C): max / mean of the point features per RoI voxel."""
    if rois.shape[1] != 7 or pts.shape[1] != 3:
        raise AssertionError("RoIAwarePool3d: rois (N, 7), pts (P, 3)")
    ox, oy, oz = (int(out_size),) * 3 if isinstance(out_size, int) else (int(v) for v in out_size)
    n, p, c = rois.shape[0], pts.shape[0], pts_feature.shape[-1]
    pooled = torch.zeros((n, ox, oy, oz, c), dtype=pts_feature.dtype, device=pts_feature.device)
    argmax = torch.zeros((n, ox, oy, oz, c), dtype=torch.int32, device=pts_feature.device)
    members = torch.zeros((n, ox, oy, oz, max_pts_each_voxel), dtype=torch.int32, device=pts_feature.device)
    code = _POOL_CODE[pool_method]
    G.run("fv2p_roiaware_pool3d_fwd", rois.contiguous(), pts.contiguous(), pts_feature.contiguous(), n, p, c, max_pts_each_voxel,
          ox, oy, oz, code, argmax, members, pooled)
    saved.update(members=members, argmax=argmax, code=code, dims=(n, ox, oy, oz, c, p, max_pts_each_voxel))
    return pooled


def _pool_grad(saved, grad):
    n, ox, oy, oz, c, p, cap = saved["dims"]
    g = torch.zeros((p, c), dtype=grad.dtype, device=grad.device)
    G.run("fv2p_roiaware_pool3d_bwd", saved["members"], saved["argmax"], grad.contiguous(), n, ox, oy, oz, c, cap, saved["code"], g)
    return None, None, g


RoIAwarePool3dFunction = G.autograd_op("RoIAwarePool3dFunction", _pool, _pool_grad)


class RoIAwarePool3d(nn.Module):
    def __init__(self, out_size, max_pts_each_voxel=128):
        super().__init__()
        self.out_size, self.max_pts_each_voxel = out_size, max_pts_each_voxel

    def forward(self, rois, pts, pts_feature, pool_method='max'):
        if pool_method not in _POOL_CODE:
            raise AssertionError("pool_method must be 'max' or 'avg'")
        return RoIAwarePool3dFunction.apply(rois, pts, pts_feature, self.out_size, self.max_pts_each_voxel, pool_method)
