"""`pcdet.ops.roipoint_pool3d.roipoint_pool3d_cuda.forward` (reference roipoint_pool3d.cpp:22-56)."""
import torch

import fv2p_native as _nat


def forward(xyz, boxes3d, pts_feature, pooled_features, pooled_empty_flag):
    """xyz (B,N,3), boxes3d (B,M,7), pts_feature (B,N,C), pooled_features (B,M,S,3+C), pooled_empty_flag (B,M) int32."""
    _nat.require_cuda(xyz, boxes3d, pts_feature, pooled_features, pooled_empty_flag)
    for t in (xyz, boxes3d, pts_feature, pooled_features, pooled_empty_flag):
        if not t.is_contiguous():
            raise _nat.Fv2pError("roipoint_pool3d: tensors must be contiguous")
    b, n, _ = xyz.shape
    m, c, s = boxes3d.shape[1], pts_feature.shape[2], pooled_features.shape[2]
    with _nat.device_guard(xyz.device):
        _nat.call("fv2p_roipoint_pool3d", xyz, boxes3d, pts_feature, b, n, m, c, s, pooled_features, pooled_empty_flag, _nat.stream())
    return 1
