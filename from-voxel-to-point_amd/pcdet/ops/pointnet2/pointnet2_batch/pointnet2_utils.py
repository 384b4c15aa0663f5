"""Batched pointnet2 operators behind the names of the reference's
pcdet/ops/pointnet2/pointnet2_batch/pointnet2_utils.py:10-380 (FurthestPointSampling, GatherOperation, ThreeNN, ThreeInterpolate,
GroupingOperation, BallQuery and their `.apply` aliases, QueryAndGroup, GroupAll, top3_interpolate[_with_grad]).

Each operator is a forward / backward function pair calling the C ABI directly (pcdet/ops/_glue.py); the ext-module name
`pointnet2_batch_cuda` with the reference's wrapper signatures lives next to this file for third-party callers."""
import torch
import torch.nn as nn

from ... import _glue as G


# ---- sampling / gathering ------------------------------------------------------------------------------------------------------
def _fps(saved, xyz, npoint):
    """xyz (B, N, 3) -> (B, npoint) int32: index 0 first, then the farthest remaining point each round."""
    b, n, _ = xyz.shape
    idx = G.new(xyz, (b, npoint), torch.int32)
    running = G.new(xyz, (b, n), fill=1e10)
    ws = G.scratch("fv2p_furthest_point_sampling_ws_bytes", xyz.device, b, n)
    G.run("fv2p_furthest_point_sampling", b, n, npoint, xyz, running, idx, ws, ws.numel())
    return idx


def _gather(saved, features, idx):
    """features (B, C, N), idx (B, M) -> (B, C, M)."""
    b, c, n = features.shape
    m = idx.shape[1]
    out = G.new(features, (b, c, m))
    G.run("fv2p_gather_points", b, c, n, m, features, idx, out)
    saved.update(idx=idx, shape=(b, c, n, m))
    return out


def _gather_grad(saved, grad):
    b, c, n, m = saved["shape"]
    g = torch.zeros((b, c, n), dtype=torch.float32, device=grad.device)
    G.run("fv2p_gather_points_grad", b, c, n, m, grad.contiguous(), saved["idx"], g)
    return g


# ---- nearest neighbours and interpolation -------------------------------------------------------------------------------------------
def _three_nn(saved, unknown, known):
    """unknown (B, N, 3), known (B, M, 3) -> (distances (B, N, 3), indices (B, N, 3)) of the three nearest known points."""
    b, n, _ = unknown.shape
    d2 = G.new(unknown, (b, n, 3))
    idx = G.new(unknown, (b, n, 3), torch.int32)
    G.run("fv2p_three_nn_batch", b, n, known.shape[1], unknown, known, d2, idx)
    return d2.sqrt(), idx


def _interp(saved, features, idx, weight):
    """features (B, C, M), idx / weight (B, N, 3) -> (B, C, N)."""
    b, c, m = features.shape
    n = idx.shape[1]
    out = G.new(features, (b, c, n))
    G.run("fv2p_three_interpolate_batch", b, c, m, n, features, idx, weight, out)
    saved.update(idx=idx, weight=weight, shape=(b, c, m, n))
    return out


def _interp_grad(saved, grad):
    b, c, m, n = saved["shape"]
    g = torch.zeros((b, c, m), dtype=torch.float32, device=grad.device)
    G.run("fv2p_three_interpolate_batch_grad", b, c, n, m, grad.contiguous(), saved["idx"], saved["weight"], g)
    return g


# ---- ball query and grouping ----------------------------------------------------------------------------------------------------------
def _ball(saved, radius, nsample, xyz, new_xyz):
    """xyz (B, N, 3), new_xyz (B, M, 3) -> (B, M, nsample) int32, the first nsample points within `radius` in index order."""
    b, n, _ = xyz.shape
    m = new_xyz.shape[1]
    idx = torch.zeros((b, m, nsample), dtype=torch.int32, device=xyz.device)
    G.run("fv2p_ball_query_batch", b, n, m, float(radius), nsample, new_xyz, xyz, idx)
    return idx


def _group(saved, features, idx):
    """features (B, C, N), idx (B, M, S) -> (B, C, M, S)."""
    b, c, n = features.shape
    _, m, s = idx.shape
    out = G.new(features, (b, c, m, s))
    G.run("fv2p_group_points_batch", b, c, n, m, s, features, idx, out)
    saved.update(idx=idx, shape=(b, c, n, m, s))
    return out


def _group_grad(saved, grad):
    b, c, n, m, s = saved["shape"]
    g = torch.zeros((b, c, n), dtype=torch.float32, device=grad.device)
    G.run("fv2p_group_points_batch_grad", b, c, n, m, s, grad.contiguous(), saved["idx"], g)
    return g


FurthestPointSampling = G.autograd_op("FurthestPointSampling", _fps)
GatherOperation = G.autograd_op("GatherOperation", _gather, _gather_grad)
ThreeNN = G.autograd_op("ThreeNN", _three_nn)
ThreeInterpolate = G.autograd_op("ThreeInterpolate", _interp, _interp_grad)
GroupingOperation = G.autograd_op("GroupingOperation", _group, _group_grad)
BallQuery = G.autograd_op("BallQuery", _ball)
furthest_point_sample, gather_operation = FurthestPointSampling.apply, GatherOperation.apply
three_nn, three_interpolate = ThreeNN.apply, ThreeInterpolate.apply
grouping_operation, ball_query = GroupingOperation.apply, BallQuery.apply


class QueryAndGroup(nn.Module):
    """Ball query around new_xyz, then the neighbours' centred coordinates (and features) gathered into (B, 3 + C, M, S)."""

    def __init__(self, radius: float, nsample: int, use_xyz: bool = True):
        super().__init__()
        self.radius, self.nsample, self.use_xyz = radius, nsample, use_xyz

    def forward(self, xyz, new_xyz, features=None):
        idx = ball_query(self.radius, self.nsample, xyz, new_xyz)
        rel = grouping_operation(xyz.transpose(1, 2).contiguous(), idx) - new_xyz.transpose(1, 2).unsqueeze(-1)
        if features is None:
            assert self.use_xyz, "Cannot have not features and not use xyz as a feature!"
            return rel
        grouped = grouping_operation(features, idx)
        return torch.cat([rel, grouped], dim=1) if self.use_xyz else grouped


class GroupAll(nn.Module):
    """One group holding every point: (B, 3 + C, 1, N)."""

    def __init__(self, use_xyz: bool = True):
        super().__init__()
        self.use_xyz = use_xyz

    def forward(self, xyz, new_xyz, features=None):
        coords = xyz.transpose(1, 2).unsqueeze(2)
        if features is None:
            return coords
        return torch.cat([coords, features.unsqueeze(2)], dim=1) if self.use_xyz else features.unsqueeze(2)


def _inverse_distance_weights(dist):
    w = 1.0 / (dist + 1e-8)
    return w / w.sum(dim=-1, keepdim=True)


def top3_interpolate(xyz, new_xyz, feats, nsamples=None):
    """Features feats (N, C) at xyz (N, 3) interpolated onto new_xyz (M, 3) from the three nearest sources with
    inverse-distance weights -> (M, C); the gradient reaches feats only (Voxel-to-Point decoder, reference :292-326)."""
    if not (xyz.dim() == new_xyz.dim() == feats.dim() == 2):
        raise NotImplementedError
    dist, idx = three_nn(new_xyz.unsqueeze(0).contiguous(), xyz.unsqueeze(0).contiguous())
    out = three_interpolate(feats.t().unsqueeze(0).contiguous(), idx, _inverse_distance_weights(dist))
    return out[0].t()


def top3_interpolate_with_grad(xyz, new_xyz, feats, nsamples=None):
    """Same interpolation with the weights recomputed from gathered coordinates, so that gradients also reach xyz and
    new_xyz (reference :329-380)."""
    if not (xyz.dim() == new_xyz.dim() == feats.dim() == 2):
        raise NotImplementedError
    _, idx = three_nn(new_xyz.detach().unsqueeze(0).contiguous(), xyz.unsqueeze(0).contiguous())
    near_xyz = grouping_operation(xyz.t().unsqueeze(0).contiguous(), idx)[0].permute(1, 2, 0)        # (M, 3, 3)
    near_feats = grouping_operation(feats.t().unsqueeze(0).contiguous(), idx)[0].permute(1, 2, 0)    # (M, 3, C)
    weight = _inverse_distance_weights((near_xyz - new_xyz.unsqueeze(1)).norm(dim=-1))
    return (near_feats * weight.unsqueeze(-1)).sum(dim=1)
