"""`pcdet.ops.roiaware_pool3d.roiaware_pool3d_cuda` — the four functions bound by the reference
(pcdet/ops/roiaware_pool3d/src/roiaware_pool3d.cpp:172-177); caller-allocated outputs are filled in place."""
import torch

import fv2p_native as _nat


def forward(rois, pts, pts_feature, argmax, pts_idx_of_voxels, pooled_features, pool_method):
    _nat.require_cuda(rois, pts, pts_feature, argmax, pts_idx_of_voxels, pooled_features)
    boxes_num, out_x, out_y, out_z, channels = pooled_features.shape
    max_pts = pts_idx_of_voxels.shape[4]
    with _nat.device_guard(rois.device):
        _nat.call("fv2p_roiaware_pool3d_fwd", rois.contiguous(), pts.contiguous(), pts_feature.contiguous(), boxes_num, pts.shape[0],
                  channels, max_pts, out_x, out_y, out_z, int(pool_method), argmax, pts_idx_of_voxels, pooled_features, _nat.stream())
    return 1


def backward(pts_idx_of_voxels, argmax, grad_out, grad_in, pool_method):
    _nat.require_cuda(pts_idx_of_voxels, argmax, grad_out, grad_in)
    boxes_num, out_x, out_y, out_z, max_pts = pts_idx_of_voxels.shape
    channels = grad_out.shape[4]
    with _nat.device_guard(grad_out.device):
        _nat.call("fv2p_roiaware_pool3d_bwd", pts_idx_of_voxels, argmax, grad_out.contiguous(), boxes_num, out_x, out_y, out_z, channels,
                  max_pts, int(pool_method), grad_in, _nat.stream())
    return 1


def points_in_boxes_gpu(boxes_tensor, pts_tensor, box_idx_of_points_tensor):
    _nat.require_cuda(boxes_tensor, pts_tensor, box_idx_of_points_tensor)
    b, t, _ = boxes_tensor.shape
    with _nat.device_guard(boxes_tensor.device):
        _nat.call("fv2p_points_in_boxes", boxes_tensor.contiguous(), pts_tensor.contiguous(), b, t, pts_tensor.shape[1],
                  box_idx_of_points_tensor, _nat.stream())
    return 1


def points_in_boxes_cpu(boxes_tensor, pts_tensor, pts_indices_tensor):
    for t in (boxes_tensor, pts_tensor, pts_indices_tensor):
        if t.is_cuda or not t.is_contiguous():
            raise _nat.Fv2pError("points_in_boxes_cpu takes contiguous CPU tensors")
    _nat.call("fv2p_points_in_boxes_cpu", boxes_tensor, pts_tensor, boxes_tensor.shape[0], pts_tensor.shape[0], pts_indices_tensor)
    return 1
