"""Set-abstraction / feature-propagation modules on the batched pointnet2 ops — class names, constructor keywords
and parameter names (`groupers`, `mlps`, `mlps_gate`, `mlp`) of the reference's
pcdet/ops/pointnet2/pointnet2_batch/pointnet2_modules.py:10-280 so checkpoints and configs carry over."""
from typing import List

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import pointnet2_utils


def _pool(x, method):
    if method == 'max_pool':
        return F.max_pool2d(x, kernel_size=[1, x.size(3)])
    if method == 'avg_pool':
        return F.avg_pool2d(x, kernel_size=[1, x.size(3)])
    raise NotImplementedError


def _sample_centres(xyz, npoint):
    if npoint is None:
        return None
    xyz_flipped = xyz.transpose(1, 2).contiguous()
    idx = pointnet2_utils.furthest_point_sample(xyz, npoint)
    return pointnet2_utils.gather_operation(xyz_flipped, idx).transpose(1, 2).contiguous()


class _PointnetSAModuleBase(nn.Module):

    def __init__(self):
        super().__init__()
        self.npoint = None
        self.groupers = None
        self.mlps = None
        self.pool_method = 'max_pool'

    def forward(self, xyz: torch.Tensor, features: torch.Tensor = None, new_xyz=None) -> (torch.Tensor, torch.Tensor):
        """xyz (B,N,3), features (B,C,N) -> new_xyz (B,npoint,3), new_features (B, sum_k mlps[k][-1], npoint)."""
        if new_xyz is None:
            new_xyz = _sample_centres(xyz, self.npoint)
        outs = []
        for grouper, mlp in zip(self.groupers, self.mlps):
            f = mlp(grouper(xyz, new_xyz, features).contiguous())
            outs.append(_pool(f, self.pool_method).squeeze(-1))
        return new_xyz, torch.cat(outs, dim=1)


class PointnetSAModuleMSG(_PointnetSAModuleBase):
    """Pointnet set abstraction layer with multiscale grouping"""

    def __init__(self, *, npoint: int, radii: List[float], nsamples: List[int], mlps: List[List[int]], bn: bool = True,
                 use_xyz: bool = True, pool_method='max_pool'):
        super().__init__()
        assert len(radii) == len(nsamples) == len(mlps)
        self.npoint = npoint
        self.groupers = nn.ModuleList()
        self.mlps = nn.ModuleList()
        for radius, nsample, mlp_spec in zip(radii, nsamples, mlps):
            self.groupers.append(pointnet2_utils.QueryAndGroup(radius, nsample, use_xyz=use_xyz)
                                 if npoint is not None else pointnet2_utils.GroupAll(use_xyz))
            if use_xyz:
                mlp_spec[0] += 3  # in place, like the reference (:80-81)
            layers = []
            for k in range(len(mlp_spec) - 1):
                layers.append(nn.Conv2d(mlp_spec[k], mlp_spec[k + 1], kernel_size=1, bias=False))
                if bn:
                    layers.append(nn.BatchNorm2d(mlp_spec[k + 1]))
                layers.append(nn.ReLU())
            self.mlps.append(nn.Sequential(*layers))
        self.pool_method = pool_method


class PointnetSAModule(PointnetSAModuleMSG):
    """Pointnet set abstraction layer"""

    def __init__(self, *, mlp: List[int], npoint: int = None, radius: float = None, nsample: int = None,
                 bn: bool = True, use_xyz: bool = True, pool_method='max_pool'):
        super().__init__(mlps=[mlp], npoint=npoint, radii=[radius], nsamples=[nsample], bn=bn, use_xyz=use_xyz,
                         pool_method=pool_method)


class PointnetFPModule(nn.Module):
    r"""Propigates the features of one set to another"""

    def __init__(self, *, mlp: List[int], bn: bool = True):
        super().__init__()
        layers = []
        for k in range(len(mlp) - 1):
            layers.extend([nn.Conv2d(mlp[k], mlp[k + 1], kernel_size=1, bias=False), nn.BatchNorm2d(mlp[k + 1]), nn.ReLU()])
        self.mlp = nn.Sequential(*layers)

    def forward(self, unknown: torch.Tensor, known: torch.Tensor, unknow_feats: torch.Tensor,
                known_feats: torch.Tensor) -> torch.Tensor:
        """unknown (B,n,3), known (B,m,3), unknow_feats (B,C1,n), known_feats (B,C2,m) -> (B, mlp[-1], n)."""
        if known is not None:
            dist, idx = pointnet2_utils.three_nn(unknown, known)
            dist_recip = 1.0 / (dist + 1e-8)
            norm = torch.sum(dist_recip, dim=2, keepdim=True)
            weight = dist_recip / norm
            interpolated_feats = pointnet2_utils.three_interpolate(known_feats, idx, weight).contiguous()
        else:
            interpolated_feats = known_feats.expand(*known_feats.size()[0:2], unknown.size(1)).contiguous()
        new_features = torch.cat([interpolated_feats, unknow_feats], dim=1) if unknow_feats is not None else interpolated_feats
        return self.mlp(new_features.unsqueeze(-1).contiguous()).squeeze(-1)


class PointnetSAModuleMSGGated(nn.Module):
    """Multiscale grouping with a sigmoid gate branch per scale (fork addition, reference :172-276)."""

    def __init__(self, *, npoint: int, radii: List[float], nsamples: List[int], mlps: List[List[int]], bn: bool = True,
                 use_xyz: bool = True, pool_method='max_pool'):
        super().__init__()
        assert len(radii) == len(nsamples) == len(mlps)
        self.npoint = npoint
        self.groupers = nn.ModuleList()
        self.mlps = nn.ModuleList()
        self.mlps_gate = nn.ModuleList()
        for radius, nsample, mlp_spec in zip(radii, nsamples, mlps):
            self.groupers.append(pointnet2_utils.QueryAndGroup(radius, nsample, use_xyz=use_xyz)
                                 if npoint is not None else pointnet2_utils.GroupAll(use_xyz))
            if use_xyz:
                mlp_spec[0] += 3
            main, gate = [], []
            for k in range(len(mlp_spec) - 1):
                main.extend([nn.Conv2d(mlp_spec[k], mlp_spec[k + 1], kernel_size=1, bias=False), nn.ReLU()])
                gate.append(nn.Conv2d(mlp_spec[k], mlp_spec[k + 1], kernel_size=1, bias=False))
            self.mlps.append(nn.Sequential(*main))
            self.mlps_gate.append(nn.Sequential(*gate))
        self.pool_method = pool_method
        self.init_weights()

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            if isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1.0)
                nn.init.constant_(m.bias, 0)

    def forward(self, xyz: torch.Tensor, features: torch.Tensor = None, new_xyz=None) -> (torch.Tensor, torch.Tensor):
        if new_xyz is None:
            new_xyz = _sample_centres(xyz, self.npoint)
        outs = []
        for grouper, mlp, gate in zip(self.groupers, self.mlps, self.mlps_gate):
            g = grouper(xyz, new_xyz, features).contiguous()
            f = mlp(g) * torch.sigmoid(gate(g))
            outs.append(_pool(f, self.pool_method).squeeze(-1))
        return new_xyz, torch.cat(outs, dim=1)
