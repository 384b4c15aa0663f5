"""Set-abstraction / feature-propagation modules on the batched pointnet2 ops — class names, constructor keywords
and parameter names (`groupers`, `mlps`, `mlps_gate`, `mlp`) of the reference's
pcdet/ops/pointnet2/pointnet2_batch/pointnet2_modules.py:10-280 so checkpoints and configs carry over."""
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


def _scales(npoint, radii, nsamples, mlps, use_xyz):
    """One (grouper, channel widths) per scale.  With use_xyz the first width grows by 3 IN the caller's list, as in the reference
    (:80-81) — configs that reuse the list see the same numbers here as there."""
    assert len(radii) == len(nsamples) == len(mlps)
    for radius, nsample, widths in zip(radii, nsamples, mlps):
        grouper = pointnet2_utils.QueryAndGroup(radius, nsample, use_xyz=use_xyz) if npoint is not None else pointnet2_utils.GroupAll(use_xyz)
        if use_xyz:
            widths[0] += 3
        yield grouper, widths


def _shared_mlp(widths, bn, relu=True):
    """1x1 Conv2d (bias-free) [+ BatchNorm2d] [+ ReLU] per consecutive pair of widths: indices 0, 1(, 2) ... of the Sequential."""
    layers = []
    for cin, cout in zip(widths[:-1], widths[1:]):
        layers.append(nn.Conv2d(cin, cout, kernel_size=1, bias=False))
        if bn:
            layers.append(nn.BatchNorm2d(cout))
        if relu:
            layers.append(nn.ReLU())
    return nn.Sequential(*layers)


class _PointnetSAModuleBase(nn.Module):

    def __init__(self):
        super().__init__()
        self.npoint, self.groupers, self.mlps, self.pool_method = None, None, None, 'max_pool'

    def forward(self, xyz, features=None, new_xyz=None):
        """xyz (B,N,3), features (B,C,N) -> new_xyz (B,npoint,3), new_features (B, sum_k mlps[k][-1], npoint)."""
        if new_xyz is None:
            new_xyz = _sample_centres(xyz, self.npoint)
        outs = []
        for grouper, mlp in zip(self.groupers, self.mlps):
            f = mlp(grouper(xyz, new_xyz, features).contiguous())
            outs.append(_pool(f, self.pool_method).squeeze(-1))
        return new_xyz, torch.cat(outs, dim=1)


class PointnetSAModuleMSG(_PointnetSAModuleBase):
    """Set abstraction with multi-scale grouping: per scale a ball query + grouping, a shared MLP and a pool over the samples."""

    def __init__(self, *, npoint, radii, nsamples, mlps, bn=True, use_xyz=True, pool_method='max_pool'):
        super().__init__()
        self.npoint, self.pool_method = npoint, pool_method
        self.groupers, self.mlps = nn.ModuleList(), nn.ModuleList()
        for grouper, widths in _scales(npoint, radii, nsamples, mlps, use_xyz):
            self.groupers.append(grouper)
            self.mlps.append(_shared_mlp(widths, bn))


class PointnetSAModule(PointnetSAModuleMSG):
    """Single-scale set abstraction."""

    def __init__(self, *, mlp, npoint=None, radius=None, nsample=None, bn=True, use_xyz=True, pool_method='max_pool'):
        super().__init__(mlps=[mlp], npoint=npoint, radii=[radius], nsamples=[nsample], bn=bn, use_xyz=use_xyz, pool_method=pool_method)


class PointnetFPModule(nn.Module):
    """Feature propagation: inverse-distance interpolation from the three nearest known points, concatenated with the skip features."""

    def __init__(self, *, mlp, bn=True):
        super().__init__()
        self.mlp = _shared_mlp(mlp, True)      # the reference always builds the BatchNorm here (:132-137)

    def forward(self, unknown, known, unknow_feats, known_feats):
        """unknown (B,n,3), known (B,m,3), unknow_feats (B,C1,n), known_feats (B,C2,m) -> (B, mlp[-1], n)."""
        if known is None:
            spread = known_feats.expand(*known_feats.size()[0:2], unknown.size(1)).contiguous()
        else:
            dist, idx = pointnet2_utils.three_nn(unknown, known)
            closeness = 1.0 / (dist + 1e-8)
            spread = pointnet2_utils.three_interpolate(known_feats, idx, closeness / closeness.sum(dim=2, keepdim=True)).contiguous()
        stacked = spread if unknow_feats is None else torch.cat([spread, unknow_feats], dim=1)
        return self.mlp(stacked.unsqueeze(-1).contiguous()).squeeze(-1)


class PointnetSAModuleMSGGated(nn.Module):
    """Multi-scale grouping with a sigmoid gate branch per scale (fork addition, reference :172-276): mlps[k] is Conv2d + ReLU per
    layer, mlps_gate[k] the bare Conv2d stack, output = pool(mlps(g) * sigmoid(mlps_gate(g)))."""

    def __init__(self, *, npoint, radii, nsamples, mlps, bn=True, use_xyz=True, pool_method='max_pool'):
        super().__init__()
        self.npoint, self.pool_method = npoint, pool_method
        self.groupers, self.mlps, self.mlps_gate = nn.ModuleList(), nn.ModuleList(), nn.ModuleList()
        for grouper, widths in _scales(npoint, radii, nsamples, mlps, use_xyz):
            self.groupers.append(grouper)
            self.mlps.append(_shared_mlp(widths, False))
            self.mlps_gate.append(_shared_mlp(widths, False, relu=False))
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

    def forward(self, xyz, features=None, new_xyz=None):
        if new_xyz is None:
            new_xyz = _sample_centres(xyz, self.npoint)
        outs = []
        for grouper, mlp, gate in zip(self.groupers, self.mlps, self.mlps_gate):
            g = grouper(xyz, new_xyz, features).contiguous()
            outs.append(_pool(mlp(g) * torch.sigmoid(gate(g)), self.pool_method).squeeze(-1))
        return new_xyz, torch.cat(outs, dim=1)
