"""VoxelQuery / VoxelQueryAndGrouping — surface of the reference's
pcdet/ops/pointnet2/pointnet2_stack/voxel_query_utils.py:10-111 (Voxel R-CNN style neighbour-voxel grouping)."""
import torch
import torch.nn as nn
from torch.autograd import Function

from . import pointnet2_stack_cuda as pointnet2
from . import pointnet2_utils


class VoxelQuery(Function):

    @staticmethod
    def forward(ctx, max_range: int, radius: float, nsample: int, xyz: torch.Tensor, new_xyz: torch.Tensor,
                new_coords: torch.Tensor, point_indices: torch.Tensor):
        """new_coords (M,4) [b,z,y,x] voxel coords of the queries, point_indices (B,Z,Y,X) voxel -> point row (or -1)
        -> idx (M,nsample) GLOBAL point rows, empty_ball_mask (M)."""
        assert new_xyz.is_contiguous()
        assert xyz.is_contiguous()
        assert new_coords.is_contiguous()
        assert point_indices.is_contiguous()
        M = new_coords.shape[0]
        B, Z, Y, X = point_indices.shape
        idx = torch.zeros((M, nsample), dtype=torch.int32, device=xyz.device)
        z_range, y_range, x_range = max_range
        pointnet2.voxel_query_wrapper(M, Z, Y, X, nsample, radius, z_range, y_range, x_range, new_xyz, xyz, new_coords,
                                      point_indices, idx)
        empty_ball_mask = (idx[:, 0] == -1)
        idx[empty_ball_mask] = 0
        return idx, empty_ball_mask

    @staticmethod
    def backward(ctx, a=None):
        return None, None, None, None


voxel_query = VoxelQuery.apply


class VoxelQueryAndGrouping(nn.Module):
    def __init__(self, max_range: int, radius: float, nsample: int):
        super().__init__()
        self.max_range, self.radius, self.nsample = max_range, radius, nsample

    def forward(self, new_coords: torch.Tensor, xyz: torch.Tensor, xyz_batch_cnt: torch.Tensor, new_xyz: torch.Tensor,
                new_xyz_batch_cnt: torch.Tensor, features: torch.Tensor, voxel2point_indices: torch.Tensor):
        """-> grouped_features (M,C,nsample), grouped_xyz (M,3,nsample), empty_ball_mask (M).

        The global->sample-local index conversion below repeats the reference line for line, including its two
        successive subtractions (:84-99): with one sample per batch (every BASELINE config that reaches this code)
        both are no-ops."""
        assert xyz.shape[0] == xyz_batch_cnt.sum(), 'xyz: %s, xyz_batch_cnt: %s' % (str(xyz.shape), str(new_xyz_batch_cnt))
        assert new_coords.shape[0] == new_xyz_batch_cnt.sum(), \
            'new_coords: %s, new_xyz_batch_cnt: %s' % (str(new_coords.shape), str(new_xyz_batch_cnt))
        batch_size = xyz_batch_cnt.shape[0]
        idx1, empty_ball_mask1 = voxel_query(self.max_range, self.radius, self.nsample, xyz, new_xyz, new_coords,
                                             voxel2point_indices)
        idx1 = idx1.view(batch_size, -1, self.nsample)
        count = 0
        for bs_idx in range(batch_size):
            idx1[bs_idx] -= count
            count += xyz_batch_cnt[bs_idx]
        idx1 = idx1.view(-1, self.nsample)
        idx1[empty_ball_mask1] = 0
        count = 0
        for i in range(batch_size):
            bs_mask = new_coords[:, 0] == i
            idx1[bs_mask] -= count
            count += xyz_batch_cnt[i]
        idx1[empty_ball_mask1] = 0
        idx, empty_ball_mask = idx1, empty_ball_mask1
        grouped_xyz = pointnet2_utils.grouping_operation(xyz, xyz_batch_cnt, idx, new_xyz_batch_cnt)
        grouped_features = pointnet2_utils.grouping_operation(features, xyz_batch_cnt, idx, new_xyz_batch_cnt)
        return grouped_features, grouped_xyz, empty_ball_mask
