"""`pointnet2_batch_cuda` — the nine wrappers the reference binds (pointnet2_batch/src/pointnet2_api.cpp:10-24),
same positional arguments (ints first, then caller-allocated tensors)."""
import torch

import fv2p_native as _nat


def _go(name, dev_tensor, *args):
    _nat.require_cuda(*[a for a in args if isinstance(a, torch.Tensor)])
    for a in args:
        if isinstance(a, torch.Tensor) and not a.is_contiguous():
            raise _nat.Fv2pError(f"{name}: tensors must be contiguous")
    with _nat.device_guard(dev_tensor.device):
        _nat.call(name, *args, _nat.stream())
    return 1


def ball_query_wrapper(b, n, m, radius, nsample, new_xyz_tensor, xyz_tensor, idx_tensor):
    return _go("fv2p_ball_query_batch", idx_tensor, b, n, m, float(radius), nsample, new_xyz_tensor, xyz_tensor, idx_tensor)


def group_points_wrapper(b, c, n, npoints, nsample, points_tensor, idx_tensor, out_tensor):
    return _go("fv2p_group_points_batch", out_tensor, b, c, n, npoints, nsample, points_tensor, idx_tensor, out_tensor)


def group_points_grad_wrapper(b, c, n, npoints, nsample, grad_out_tensor, idx_tensor, grad_points_tensor):
    return _go("fv2p_group_points_batch_grad", grad_out_tensor, b, c, n, npoints, nsample, grad_out_tensor, idx_tensor, grad_points_tensor)


def gather_points_wrapper(b, c, n, npoints, points_tensor, idx_tensor, out_tensor):
    return _go("fv2p_gather_points", out_tensor, b, c, n, npoints, points_tensor, idx_tensor, out_tensor)


def gather_points_grad_wrapper(b, c, n, npoints, grad_out_tensor, idx_tensor, grad_points_tensor):
    return _go("fv2p_gather_points_grad", grad_out_tensor, b, c, n, npoints, grad_out_tensor, idx_tensor, grad_points_tensor)


def furthest_point_sampling_wrapper(b, n, m, points_tensor, temp_tensor, idx_tensor):
    with _nat.device_guard(idx_tensor.device):   # scratch of the bucketed (lazy, bit-identical) kernel
        ws = _nat.workspace(_nat.lib().fv2p_furthest_point_sampling_ws_bytes(b, n), idx_tensor.device)
    return _go("fv2p_furthest_point_sampling", idx_tensor, b, n, m, points_tensor, temp_tensor, idx_tensor, ws, ws.numel())


def three_nn_wrapper(b, n, m, unknown_tensor, known_tensor, dist2_tensor, idx_tensor):
    return _go("fv2p_three_nn_batch", idx_tensor, b, n, m, unknown_tensor, known_tensor, dist2_tensor, idx_tensor)


def three_interpolate_wrapper(b, c, m, n, points_tensor, idx_tensor, weight_tensor, out_tensor):
    return _go("fv2p_three_interpolate_batch", out_tensor, b, c, m, n, points_tensor, idx_tensor, weight_tensor, out_tensor)


def three_interpolate_grad_wrapper(b, c, n, m, grad_out_tensor, idx_tensor, weight_tensor, grad_points_tensor):
    return _go("fv2p_three_interpolate_batch_grad", grad_out_tensor, b, c, n, m, grad_out_tensor, idx_tensor, weight_tensor, grad_points_tensor)
