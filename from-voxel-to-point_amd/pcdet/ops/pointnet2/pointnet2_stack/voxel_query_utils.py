"""VoxelQuery / VoxelQueryAndGrouping behind the names of the reference's
pcdet/ops/pointnet2/pointnet2_stack/voxel_query_utils.py:10-111 (Voxel R-CNN style neighbour-voxel grouping)."""
import torch
import torch.nn as nn

from ... import _glue as G
from . import pointnet2_utils


def _voxel_query(saved, max_range, radius, nsample, xyz, new_xyz, new_coords, point_indices):
    """new_coords (M, 4) [b, z, y, x] voxel of every query, point_indices (B, Z, Y, X) voxel -> point row (or -1)
    -> (idx (M, nsample) GLOBAL point rows, empty_ball_mask (M)); empty queries come back as row 0."""
    m = new_coords.shape[0]
    _, z, y, x = point_indices.shape
    rz, ry, rx = max_range
    idx = torch.zeros((m, nsample), dtype=torch.int32, device=xyz.device)
    i32 = lambda t: t if t.dtype == torch.int32 else t.int()
    G.run("fv2p_voxel_query_stack", m, z, y, x, nsample, float(radius), rz, ry, rx, new_xyz, xyz, i32(new_coords), i32(point_indices), idx)
    empty = idx[:, 0] == -1
    idx[empty] = 0
    return idx, empty


VoxelQuery = G.autograd_op("VoxelQuery", _voxel_query)
voxel_query = VoxelQuery.apply


class VoxelQueryAndGrouping(nn.Module):
    def __init__(self, max_range: int, radius: float, nsample: int):
        super().__init__()
        self.max_range, self.radius, self.nsample = max_range, radius, nsample

    def forward(self, new_coords, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, features, voxel2point_indices):
        """-> (grouped_features (M, C, S), grouped_xyz (M, 3, S), empty_ball_mask (M)).

        The query returns global rows; grouping wants rows local to the sample.  The reference converts twice in a row
        (:84-99: once per sample block of a (B, M/B, S) view, once per batch-index mask) — with one sample per batch, the only
        case its configs reach, both are no-ops; both are kept so that multi-sample results stay what the reference gives."""
        if xyz.shape[0] != int(xyz_batch_cnt.sum()) or new_coords.shape[0] != int(new_xyz_batch_cnt.sum()):
            raise AssertionError("rows and batch counts disagree")
        idx, empty = voxel_query(self.max_range, self.radius, self.nsample, xyz, new_xyz, new_coords, voxel2point_indices)
        starts = torch.cumsum(xyz_batch_cnt, 0) - xyz_batch_cnt                                   # first row of every sample
        b = xyz_batch_cnt.shape[0]
        idx = (idx.view(b, -1, self.nsample) - starts.view(b, 1, 1).to(idx.dtype)).view(-1, self.nsample)
        idx[empty] = 0
        idx = idx - starts[new_coords[:, 0].long()].view(-1, 1).to(idx.dtype)
        idx[empty] = 0
        grouped_xyz = pointnet2_utils.grouping_operation(xyz, xyz_batch_cnt, idx.contiguous(), new_xyz_batch_cnt)
        grouped_features = pointnet2_utils.grouping_operation(features, xyz_batch_cnt, idx.contiguous(), new_xyz_batch_cnt)
        return grouped_features, grouped_xyz, empty
