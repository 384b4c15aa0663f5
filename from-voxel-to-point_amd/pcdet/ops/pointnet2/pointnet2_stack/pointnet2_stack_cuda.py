"""`pointnet2_stack_cuda` — the eight wrappers the reference binds (pointnet2_stack/src/pointnet2_api.cpp:11-23),
same positional arguments; outputs are caller-allocated and filled in place."""
import torch

import fv2p_native as _nat


def _go(name, dev_tensor, *args):
    ts = [a for a in args if isinstance(a, torch.Tensor)]
    _nat.require_cuda(*ts)
    for a in ts:
        if not a.is_contiguous():
            raise _nat.Fv2pError(f"{name}: tensors must be contiguous")
    with _nat.device_guard(dev_tensor.device):
        _nat.call(name, *args, _nat.stream())
    return 1


def _i32(t):
    return t if t.dtype == torch.int32 else t.int()


def ball_query_wrapper(B, M, radius, nsample, new_xyz, new_xyz_batch_cnt, xyz, xyz_batch_cnt, idx):
    return _go("fv2p_ball_query_stack", idx, B, M, float(radius), nsample, new_xyz, _i32(new_xyz_batch_cnt), xyz, _i32(xyz_batch_cnt), idx)


def voxel_query_wrapper(M, R1, R2, R3, nsample, radius, z_range, y_range, x_range, new_xyz, xyz, new_coords, point_indices, idx):
    return _go("fv2p_voxel_query_stack", idx, M, R1, R2, R3, nsample, float(radius), z_range, y_range, x_range, new_xyz, xyz,
               _i32(new_coords), _i32(point_indices), idx)


def furthest_point_sampling_wrapper(b, n, m, points_tensor, temp_tensor, idx_tensor):
    with _nat.device_guard(idx_tensor.device):   # scratch of the bucketed (lazy, bit-identical) kernel
        ws = _nat.workspace(_nat.lib().fv2p_furthest_point_sampling_ws_bytes(b, n), idx_tensor.device)
    return _go("fv2p_furthest_point_sampling", idx_tensor, b, n, m, points_tensor, temp_tensor, idx_tensor, ws, ws.numel())


def group_points_wrapper(B, M, C, nsample, features, features_batch_cnt, idx, idx_batch_cnt, out):
    return _go("fv2p_group_points_stack", out, B, M, C, nsample, features, _i32(features_batch_cnt), idx, _i32(idx_batch_cnt), out)


def group_points_grad_wrapper(B, M, C, N, nsample, grad_out, idx, idx_batch_cnt, features_batch_cnt, grad_features):
    return _go("fv2p_group_points_stack_grad", grad_out, B, M, C, N, nsample, grad_out, idx, _i32(idx_batch_cnt),
               _i32(features_batch_cnt), grad_features)


def three_nn_wrapper(unknown, unknown_batch_cnt, known, known_batch_cnt, dist2, idx):
    return _go("fv2p_three_nn_stack", idx, unknown_batch_cnt.shape[0], unknown.shape[0], known.shape[0], unknown,
               _i32(unknown_batch_cnt), known, _i32(known_batch_cnt), dist2, idx)


def three_interpolate_wrapper(features, idx, weight, out):
    return _go("fv2p_three_interpolate_stack", out, idx.shape[0], features.shape[1], features, idx, weight, out)


def three_interpolate_grad_wrapper(grad_out, idx, weight, grad_features):
    return _go("fv2p_three_interpolate_stack_grad", grad_out, idx.shape[0], grad_out.shape[1], grad_out, idx, weight, grad_features)
