"""StackSAModuleMSG / StackPointnetFPModule — surface of the reference's
pcdet/ops/pointnet2/pointnet2_stack/pointnet2_modules.py:10-137 (PV-RCNN style consumers of the stacked ops)."""
from typing import List

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import pointnet2_utils


def _init(module):
    for m in module.modules():
        if isinstance(m, (nn.Conv2d, nn.Conv1d)):
            nn.init.kaiming_normal_(m.weight)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        if isinstance(m, (nn.BatchNorm2d, nn.BatchNorm1d)):
            nn.init.constant_(m.weight, 1.0)
            nn.init.constant_(m.bias, 0)


def _pool(x, method):
    if method == 'max_pool':
        return F.max_pool2d(x, kernel_size=[1, x.size(3)]).squeeze(dim=-1)
    if method == 'avg_pool':
        return F.avg_pool2d(x, kernel_size=[1, x.size(3)]).squeeze(dim=-1)
    raise NotImplementedError


class StackSAModuleMSG(nn.Module):

    def __init__(self, *, radii: List[float], nsamples: List[int], mlps: List[List[int]], use_xyz: bool = True,
                 pool_method='max_pool'):
        super().__init__()
        assert len(radii) == len(nsamples) == len(mlps)
        self.groupers = nn.ModuleList()
        self.mlps = nn.ModuleList()
        for radius, nsample, mlp_spec in zip(radii, nsamples, mlps):
            self.groupers.append(pointnet2_utils.QueryAndGroup(radius, nsample, use_xyz=use_xyz))
            if use_xyz:
                mlp_spec[0] += 3
            layers = []
            for k in range(len(mlp_spec) - 1):
                layers.extend([nn.Conv2d(mlp_spec[k], mlp_spec[k + 1], kernel_size=1, bias=False),
                               nn.BatchNorm2d(mlp_spec[k + 1]), nn.ReLU()])
            self.mlps.append(nn.Sequential(*layers))
        self.pool_method = pool_method
        self.init_weights()

    def init_weights(self):
        _init(self)

    def forward(self, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, features=None, empty_voxel_set_zeros=True):
        """xyz (N,3), new_xyz (M,3), features (N,C) -> new_xyz, new_features (M, sum_k mlps[k][-1])."""
        outs = []
        for grouper, mlp in zip(self.groupers, self.mlps):
            f, _ = grouper(xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, features)  # (M, C, nsample)
            f = mlp(f.permute(1, 0, 2).unsqueeze(dim=0))                              # (1, C', M, nsample)
            outs.append(_pool(f, self.pool_method).squeeze(dim=0).permute(1, 0))
        return new_xyz, torch.cat(outs, dim=1)


class StackPointnetFPModule(nn.Module):
    def __init__(self, *, mlp: List[int]):
        super().__init__()
        layers = []
        for k in range(len(mlp) - 1):
            layers.extend([nn.Conv2d(mlp[k], mlp[k + 1], kernel_size=1, bias=False), nn.BatchNorm2d(mlp[k + 1]), nn.ReLU()])
        self.mlp = nn.Sequential(*layers)

    def forward(self, unknown, unknown_batch_cnt, known, known_batch_cnt, unknown_feats=None, known_feats=None):
        """unknown (N,3), known (M,3), known_feats (M,C2) [, unknown_feats (N,C1)] -> (N, C_out)."""
        dist, idx = pointnet2_utils.three_nn(unknown, unknown_batch_cnt, known, known_batch_cnt)
        dist_recip = 1.0 / (dist + 1e-8)
        norm = torch.sum(dist_recip, dim=-1, keepdim=True)
        weight = dist_recip / norm
        interpolated_feats = pointnet2_utils.three_interpolate(known_feats, idx, weight)
        new_features = torch.cat([interpolated_feats, unknown_feats], dim=1) if unknown_feats is not None else interpolated_feats
        new_features = self.mlp(new_features.permute(1, 0)[None, :, :, None])
        return new_features.squeeze(dim=0).squeeze(dim=-1).permute(1, 0)
