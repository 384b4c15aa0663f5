"""Batched pointnet2 autograd functions — call surface of the reference's
pcdet/ops/pointnet2/pointnet2_batch/pointnet2_utils.py:10-380 (FPS, gather, three_nn, three_interpolate, grouping,
ball_query, QueryAndGroup, GroupAll, top3_interpolate[_with_grad])."""
from typing import Tuple

import torch
import torch.nn as nn
from torch.autograd import Function

from . import pointnet2_batch_cuda as pointnet2


def _i32(*shape, device):
    return torch.empty(shape, dtype=torch.int32, device=device)


def _f32(*shape, device):
    return torch.empty(shape, dtype=torch.float32, device=device)


class FurthestPointSampling(Function):
    @staticmethod
    def forward(ctx, xyz: torch.Tensor, npoint: int) -> torch.Tensor:
        """xyz (B,N,3) -> (B,npoint) int32 indices; first index 0, then iterative farthest point."""
        assert xyz.is_contiguous()
        B, N, _ = xyz.size()
        output = _i32(B, npoint, device=xyz.device)
        temp = _f32(B, N, device=xyz.device).fill_(1e10)
        pointnet2.furthest_point_sampling_wrapper(B, N, npoint, xyz, temp, output)
        return output

    @staticmethod
    def backward(xyz, a=None):
        return None, None


furthest_point_sample = FurthestPointSampling.apply


class GatherOperation(Function):
    @staticmethod
    def forward(ctx, features: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
        """features (B,C,N), idx (B,npoint) -> (B,C,npoint)."""
        assert features.is_contiguous()
        assert idx.is_contiguous()
        B, npoint = idx.size()
        _, C, N = features.size()
        output = _f32(B, C, npoint, device=features.device)
        pointnet2.gather_points_wrapper(B, C, N, npoint, features, idx, output)
        ctx.for_backwards = (idx, C, N)
        return output

    @staticmethod
    def backward(ctx, grad_out):
        idx, C, N = ctx.for_backwards
        B, npoint = idx.size()
        grad_features = torch.zeros((B, C, N), dtype=torch.float32, device=grad_out.device)
        pointnet2.gather_points_grad_wrapper(B, C, N, npoint, grad_out.contiguous(), idx, grad_features)
        return grad_features, None


gather_operation = GatherOperation.apply


class ThreeNN(Function):
    @staticmethod
    def forward(ctx, unknown: torch.Tensor, known: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """unknown (B,N,3), known (B,M,3) -> (dist (B,N,3) l2 distances, idx (B,N,3))."""
        assert unknown.is_contiguous()
        assert known.is_contiguous()
        B, N, _ = unknown.size()
        m = known.size(1)
        dist2 = _f32(B, N, 3, device=unknown.device)
        idx = _i32(B, N, 3, device=unknown.device)
        pointnet2.three_nn_wrapper(B, N, m, unknown, known, dist2, idx)
        return torch.sqrt(dist2), idx

    @staticmethod
    def backward(ctx, a=None, b=None):
        return None, None


three_nn = ThreeNN.apply


class ThreeInterpolate(Function):
    @staticmethod
    def forward(ctx, features: torch.Tensor, idx: torch.Tensor, weight: torch.Tensor) -> torch.Tensor:
        """features (B,C,M), idx (B,N,3), weight (B,N,3) -> (B,C,N)."""
        assert features.is_contiguous()
        assert idx.is_contiguous()
        assert weight.is_contiguous()
        B, c, m = features.size()
        n = idx.size(1)
        ctx.three_interpolate_for_backward = (idx, weight, m)
        output = _f32(B, c, n, device=features.device)
        pointnet2.three_interpolate_wrapper(B, c, m, n, features, idx, weight, output)
        return output

    @staticmethod
    def backward(ctx, grad_out: torch.Tensor):
        idx, weight, m = ctx.three_interpolate_for_backward
        B, c, n = grad_out.size()
        grad_features = torch.zeros((B, c, m), dtype=torch.float32, device=grad_out.device)
        pointnet2.three_interpolate_grad_wrapper(B, c, n, m, grad_out.contiguous(), idx, weight, grad_features)
        return grad_features, None, None


three_interpolate = ThreeInterpolate.apply


class GroupingOperation(Function):
    @staticmethod
    def forward(ctx, features: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
        """features (B,C,N), idx (B,npoint,nsample) -> (B,C,npoint,nsample)."""
        assert features.is_contiguous()
        assert idx.is_contiguous()
        B, nfeatures, nsample = idx.size()
        _, C, N = features.size()
        output = _f32(B, C, nfeatures, nsample, device=features.device)
        pointnet2.group_points_wrapper(B, C, N, nfeatures, nsample, features, idx, output)
        ctx.for_backwards = (idx, N)
        return output

    @staticmethod
    def backward(ctx, grad_out: torch.Tensor):
        idx, N = ctx.for_backwards
        B, C, npoint, nsample = grad_out.size()
        grad_features = torch.zeros((B, C, N), dtype=torch.float32, device=grad_out.device)
        pointnet2.group_points_grad_wrapper(B, C, N, npoint, nsample, grad_out.contiguous(), idx, grad_features)
        return grad_features, None


grouping_operation = GroupingOperation.apply


class BallQuery(Function):
    @staticmethod
    def forward(ctx, radius: float, nsample: int, xyz: torch.Tensor, new_xyz: torch.Tensor) -> torch.Tensor:
        """xyz (B,N,3), new_xyz (B,npoint,3) -> idx (B,npoint,nsample) int32."""
        assert new_xyz.is_contiguous()
        assert xyz.is_contiguous()
        B, N, _ = xyz.size()
        npoint = new_xyz.size(1)
        idx = torch.zeros((B, npoint, nsample), dtype=torch.int32, device=xyz.device)
        pointnet2.ball_query_wrapper(B, N, npoint, radius, nsample, new_xyz, xyz, idx)
        return idx

    @staticmethod
    def backward(ctx, a=None):
        return None, None, None, None


ball_query = BallQuery.apply


class QueryAndGroup(nn.Module):
    def __init__(self, radius: float, nsample: int, use_xyz: bool = True):
        super().__init__()
        self.radius, self.nsample, self.use_xyz = radius, nsample, use_xyz

    def forward(self, xyz: torch.Tensor, new_xyz: torch.Tensor, features: torch.Tensor = None) -> Tuple[torch.Tensor]:
        """xyz (B,N,3), new_xyz (B,npoint,3), features (B,C,N) -> (B, 3+C, npoint, nsample)."""
        idx = ball_query(self.radius, self.nsample, xyz, new_xyz)
        xyz_trans = xyz.transpose(1, 2).contiguous()
        grouped_xyz = grouping_operation(xyz_trans, idx)
        grouped_xyz -= new_xyz.transpose(1, 2).unsqueeze(-1)
        if features is not None:
            grouped_features = grouping_operation(features, idx)
            new_features = torch.cat([grouped_xyz, grouped_features], dim=1) if self.use_xyz else grouped_features
        else:
            assert self.use_xyz, "Cannot have not features and not use xyz as a feature!"
            new_features = grouped_xyz
        return new_features


class GroupAll(nn.Module):
    def __init__(self, use_xyz: bool = True):
        super().__init__()
        self.use_xyz = use_xyz

    def forward(self, xyz: torch.Tensor, new_xyz: torch.Tensor, features: torch.Tensor = None):
        grouped_xyz = xyz.transpose(1, 2).unsqueeze(2)
        if features is not None:
            grouped_features = features.unsqueeze(2)
            new_features = torch.cat([grouped_xyz, grouped_features], dim=1) if self.use_xyz else grouped_features
        else:
            new_features = grouped_xyz
        return new_features


def top3_interpolate(xyz, new_xyz, feats, nsamples=None):
    """xyz (N,3) with feats (N,Cf) -> inverse-distance interpolation at new_xyz (M,3): (M,Cf)
    (Voxel-to-Point decoder, reference :292-326; gradient flows to feats only)."""
    if not (len(xyz.shape) == len(new_xyz.shape) == len(feats.shape) == 2):
        raise NotImplementedError
    xyz_batch = xyz.unsqueeze(dim=0)
    new_xyz_batch = new_xyz.unsqueeze(dim=0)
    feats_batch = feats.unsqueeze(dim=0).permute(0, 2, 1).contiguous()
    dist, idx = three_nn(new_xyz_batch.contiguous(), xyz_batch.contiguous())
    dist_recip = 1.0 / (dist + 1e-8)
    norm = torch.sum(dist_recip, dim=2, keepdim=True)
    weight = dist_recip / norm
    out = three_interpolate(feats_batch, idx, weight).contiguous()
    return out.permute(0, 2, 1).squeeze(dim=0)


def top3_interpolate_with_grad(xyz, new_xyz, feats, nsamples=None):
    """As top3_interpolate, but the weights are recomputed from gathered coordinates so that gradients also reach
    xyz / new_xyz (reference :329-380)."""
    if not (len(xyz.shape) == len(new_xyz.shape) == len(feats.shape) == 2):
        raise NotImplementedError
    xyz_batch = xyz.unsqueeze(dim=0)
    new_xyz_batch = new_xyz.detach().unsqueeze(dim=0)
    feats_batch = feats.unsqueeze(dim=0).permute(0, 2, 1).contiguous()
    trans_xyz_batch = xyz.unsqueeze(dim=0).permute(0, 2, 1).contiguous()
    _, idx = three_nn(new_xyz_batch.contiguous(), xyz_batch.contiguous())
    threenn_xyz = grouping_operation(trans_xyz_batch, idx).squeeze(dim=0).permute(1, 2, 0).contiguous()
    threenn_feats = grouping_operation(feats_batch, idx).squeeze(dim=0).permute(1, 2, 0).contiguous()
    dist = torch.norm((threenn_xyz - new_xyz.unsqueeze(dim=1).expand(-1, idx.shape[-1], -1)), dim=-1)
    dist_recip = 1.0 / (dist + 1e-8)
    norm = torch.sum(dist_recip, dim=1, keepdim=True)
    weight = dist_recip / norm
    return torch.sum(threenn_feats * weight.unsqueeze(-1), dim=1)
