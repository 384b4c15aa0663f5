"""The few helpers of the reference's pcdet/utils/common_utils.py that the ops layer itself imports
(check_numpy_to_torch :14-17, rotate_points_along_z :34-56).  When this tree is overlaid on a reference
checkout the reference's full module takes precedence."""
import numpy as np
import torch


def check_numpy_to_torch(x):
    if isinstance(x, np.ndarray):
        return torch.from_numpy(x).float(), True
    return x, False


def rotate_points_along_z(points, angle):
    """points (B, N, 3+C), angle (B) -> points rotated about z."""
    points, is_numpy = check_numpy_to_torch(points)
    angle, _ = check_numpy_to_torch(angle)
    cosa, sina = torch.cos(angle), torch.sin(angle)
    zeros, ones = angle.new_zeros(points.shape[0]), angle.new_ones(points.shape[0])
    rot = torch.stack((cosa, sina, zeros, -sina, cosa, zeros, zeros, zeros, ones), dim=1).view(-1, 3, 3).float()
    out = torch.cat((torch.matmul(points[:, :, 0:3], rot), points[:, :, 3:]), dim=-1)
    return out.numpy() if is_numpy else out
