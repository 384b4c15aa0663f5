"""NeighborVoxelSAModuleMSG — surface of the reference's
pcdet/ops/pointnet2/pointnet2_stack/voxel_pool_modules.py:8-140 (parameter names mlps_in / mlps_pos / mlps_out)."""
from typing import List

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import voxel_query_utils


class NeighborVoxelSAModuleMSG(nn.Module):

    def __init__(self, *, query_ranges: List[List[int]], radii: List[float], nsamples: List[int], mlps: List[List[int]],
                 use_xyz: bool = True, pool_method='max_pool'):
        super().__init__()
        assert len(query_ranges) == len(nsamples) == len(mlps)
        self.groupers = nn.ModuleList()
        self.mlps_in = nn.ModuleList()
        self.mlps_pos = nn.ModuleList()
        self.mlps_out = nn.ModuleList()
        for max_range, nsample, radius, spec in zip(query_ranges, nsamples, radii, mlps):
            self.groupers.append(voxel_query_utils.VoxelQueryAndGrouping(max_range, radius, nsample))
            self.mlps_in.append(nn.Sequential(nn.Conv1d(spec[0], spec[1], kernel_size=1, bias=False), nn.BatchNorm1d(spec[1])))
            self.mlps_pos.append(nn.Sequential(nn.Conv2d(3, spec[1], kernel_size=1, bias=False), nn.BatchNorm2d(spec[1])))
            self.mlps_out.append(nn.Sequential(nn.Conv1d(spec[1], spec[2], kernel_size=1, bias=False), nn.BatchNorm1d(spec[2]),
                                               nn.ReLU()))
        self.relu = nn.ReLU()
        self.pool_method = pool_method
        self.init_weights()

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, (nn.Conv2d, nn.Conv1d)):
                nn.init.kaiming_normal_(m.weight)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            if isinstance(m, (nn.BatchNorm2d, nn.BatchNorm1d)):
                nn.init.constant_(m.weight, 1.0)
                nn.init.constant_(m.bias, 0)

    def forward(self, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, new_coords, features, voxel2point_indices):
        """-> new_features (M, sum_k mlps[k][-1])."""
        new_coords = new_coords[:, [0, 3, 2, 1]].contiguous()
        outs = []
        for k in range(len(self.groupers)):
            f_in = self.mlps_in[k](features.permute(1, 0).unsqueeze(0)).permute(0, 2, 1).contiguous()
            f_in = f_in.view(-1, f_in.shape[-1])
            gf, gxyz, empty = self.groupers[k](new_coords, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, f_in,
                                               voxel2point_indices)
            gf[empty] = 0
            gf = gf.permute(1, 0, 2).unsqueeze(dim=0)
            gxyz = gxyz - new_xyz.unsqueeze(-1)
            gxyz[empty] = 0
            pos = self.mlps_pos[k](gxyz.permute(1, 0, 2).unsqueeze(0))
            f = self.relu(gf + pos)
            if self.pool_method == 'max_pool':
                f = F.max_pool2d(f, kernel_size=[1, f.size(3)]).squeeze(dim=-1)
            elif self.pool_method == 'avg_pool':
                f = F.avg_pool2d(f, kernel_size=[1, f.size(3)]).squeeze(dim=-1)
            else:
                raise NotImplementedError
            outs.append(self.mlps_out[k](f).squeeze(dim=0).permute(1, 0))
        return torch.cat(outs, dim=1)
