"""Stacked-batch pointnet2 operators behind the names of the reference's
pcdet/ops/pointnet2/pointnet2_stack/pointnet2_utils.py:8-262.  Rows of all samples are concatenated; `*_batch_cnt`
(int32 [B]) holds the rows per sample.  Forward / backward function pairs on the C ABI (pcdet/ops/_glue.py)."""
import os

import torch
import torch.nn as nn

from ... import _glue as G


def _cnt(t):
    return (t if t.dtype == torch.int32 else t.int()).contiguous()


def _ball(saved, radius, nsample, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt):
    """-> (idx (M, nsample) rows LOCAL to the sample, empty_ball_mask (M) bool); empty balls come back as row 0."""
    m = new_xyz.shape[0]
    idx = torch.zeros((m, nsample), dtype=torch.int32, device=xyz.device)
    G.run("fv2p_ball_query_stack", xyz_batch_cnt.shape[0], m, float(radius), nsample, new_xyz, _cnt(new_xyz_batch_cnt), xyz,
          _cnt(xyz_batch_cnt), idx)
    empty = idx[:, 0] == -1
    idx[empty] = 0
    return idx, empty


def _group(saved, features, features_batch_cnt, idx, idx_batch_cnt):
    """features (N, C), idx (M, S) sample-local -> (M, C, S)."""
    if features.shape[0] != int(features_batch_cnt.sum()) or idx.shape[0] != int(idx_batch_cnt.sum()):
        raise AssertionError(f"rows and batch counts disagree: features {tuple(features.shape)} / {features_batch_cnt.tolist()}, "
                             f"idx {tuple(idx.shape)} / {idx_batch_cnt.tolist()}")
    (m, s), (n, c) = idx.shape, features.shape
    b = idx_batch_cnt.shape[0]
    out = G.new(features, (m, c, s))
    G.run("fv2p_group_points_stack", b, m, c, s, features, _cnt(features_batch_cnt), idx, _cnt(idx_batch_cnt), out)
    saved.update(idx=idx, fc=_cnt(features_batch_cnt), ic=_cnt(idx_batch_cnt), dims=(b, m, c, n, s))
    return out


def _group_grad(saved, grad):
    b, m, c, n, s = saved["dims"]
    g = torch.zeros((n, c), dtype=torch.float32, device=grad.device)
    G.run("fv2p_group_points_stack_grad", b, m, c, n, s, grad.contiguous(), saved["idx"], saved["ic"], saved["fc"], g)
    return g


def _fps(saved, xyz, npoint):
    """xyz (B, N, 3) -> (B, npoint) int32 (the V2P decoder calls it once per sample, residual_v2p_decoder.py:216)."""
    b, n, _ = xyz.shape
    idx = G.new(xyz, (b, npoint), torch.int32)
    running = G.new(xyz, (b, n), fill=1e10)
    ws = G.scratch("fv2p_furthest_point_sampling_ws_bytes", xyz.device, b, n)
    G.run("fv2p_furthest_point_sampling", b, n, npoint, xyz, running, idx, ws, ws.numel())
    return idx


# known points PER SAMPLE below which the tiled scan stays ahead of building a grid (measured, profiles/r03_microbench_nn.txt: at the
# KITTI levels, 5 - 35 k voxel centres over four samples for 49 152 queries, the scan takes 84 - 492 us and the grid 170 - 1 850 us — the
# scan's cost goes with the sample's own points, the grid pays nine launches and 2 M cells to clear and scan; at the Waymo levels,
# 24 - 110 k centres in one sample, the grid takes 108 - 419 us against 339 - 1 525 us)
GRID_MIN_KNOWN = 3000


def _three_nn(saved, unknown, unknown_batch_cnt, known, known_batch_cnt, cell=None):
    """unknown (N, 3), known (M, 3) -> (distances (N, 3), GLOBAL rows into known (N, 3)).
    cell (optional, not in the reference's signature): spacing hint for the grid search (fv2p_three_nn_stack_grid) that answers
    large known sets — same result as the scan, bit for bit; None lets the library estimate it."""
    if unknown.dim() != 2 or unknown.shape[1] != 3 or known.dim() != 2 or known.shape[1] != 3 or len(unknown_batch_cnt) != len(known_batch_cnt):
        raise AssertionError("three_nn (stack): expects (N, 3) / (M, 3) and one count per sample on both sides")
    d2 = torch.zeros_like(unknown)
    idx = torch.zeros(unknown.shape, dtype=torch.int32, device=unknown.device)
    nb = len(unknown_batch_cnt)
    if known.shape[0] >= GRID_MIN_KNOWN * nb and nb <= 512 and os.environ.get("FV2P_NN_GRID", "1") != "0":
        ws = G.scratch("fv2p_three_nn_grid_ws_bytes", unknown.device, nb, unknown.shape[0], known.shape[0])
        G.run("fv2p_three_nn_stack_grid", nb, unknown.shape[0], known.shape[0], unknown.contiguous(), _cnt(unknown_batch_cnt),
              known.contiguous(), _cnt(known_batch_cnt), float(cell) if cell else 0.0, d2, idx, ws, ws.numel())
    else:
        G.run("fv2p_three_nn_stack", nb, unknown.shape[0], known.shape[0], unknown.contiguous(), _cnt(unknown_batch_cnt),
              known.contiguous(), _cnt(known_batch_cnt), d2, idx)
    return d2.sqrt(), idx


def _interp(saved, features, idx, weight):
    """features (M, C), idx / weight (N, 3) -> (N, C)."""
    if idx.shape != weight.shape or idx.shape[1] != 3:
        raise AssertionError("three_interpolate (stack): idx and weight must both be (N, 3)")
    idx, weight = idx.contiguous(), weight.contiguous()
    out = torch.zeros((idx.shape[0], features.shape[1]), dtype=features.dtype, device=features.device)
    G.run("fv2p_three_interpolate_stack", idx.shape[0], features.shape[1], features.contiguous(), idx, weight, out)
    saved.update(idx=idx, weight=weight, rows=features.shape[0])
    return out


# The interpolation gradient from 8192 queries on is the library's gather form (fv2p_three_interpolate_stack_grad_gather: (row, entry)
# keys radix-sorted, the sorted sequence summed in 32-entry segments, runs that cross segments closed from the partial sums in segment
# order): no float atomics, no caller-side zero fill, BIT-IDENTICAL from run to run.  profiles/r04_op_roofline.txt: 139 / 155 / 157 us at
# C = 16 / 64 / 128 against 115 / 250 / 240 us of the scatter form on the microbench's random known points (a few rows hold thousands of
# entries there: the atomics contend); on the decoder's own lists, where no row is hot, it is the sort that costs (~40 of ~80 us, against
# ~50 us scatter): +0.5 ms on the 49 ms in-line step, nothing on the scheduled one.  FV2P_INTERP_GATHER=0 selects the scatter form (zero
# fill + float atomics, the reference's interpolate_gpu.cu:105-160).
GATHER_GRAD_MIN_QUERIES = 8192


def _interp_grad(saved, grad):
    n, c, m = grad.shape[0], grad.shape[1], saved["rows"]
    if n >= GATHER_GRAD_MIN_QUERIES and os.environ.get("FV2P_INTERP_GATHER", "1") != "0":
        g = torch.empty((m, c), dtype=grad.dtype, device=grad.device)
        ws = G.scratch("fv2p_three_interpolate_stack_grad_ws_bytes", grad.device, n, c, m)
        G.run("fv2p_three_interpolate_stack_grad_gather", n, c, m, grad.contiguous(), saved["idx"], saved["weight"], g, ws, ws.numel())
        return g
    g = torch.zeros((m, c), dtype=grad.dtype, device=grad.device)
    G.run("fv2p_three_interpolate_stack_grad", n, c, grad.contiguous(), saved["idx"], saved["weight"], g)
    return g


BallQuery = G.autograd_op("BallQuery", _ball)
GroupingOperation = G.autograd_op("GroupingOperation", _group, _group_grad)
FurthestPointSampling = G.autograd_op("FurthestPointSampling", _fps)
ThreeNN = G.autograd_op("ThreeNN", _three_nn)
ThreeInterpolate = G.autograd_op("ThreeInterpolate", _interp, _interp_grad)
ball_query, grouping_operation = BallQuery.apply, GroupingOperation.apply
furthest_point_sample = FurthestPointSampling.apply
three_nn, three_interpolate = ThreeNN.apply, ThreeInterpolate.apply


class QueryAndGroup(nn.Module):
    """-> (new_features (M, 3 + C, S), idx (M, S)): centred neighbour coordinates and features, zeros for empty balls."""

    def __init__(self, radius: float, nsample: int, use_xyz: bool = True):
        super().__init__()
        self.radius, self.nsample, self.use_xyz = radius, nsample, use_xyz

    def forward(self, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, features=None):
        if xyz.shape[0] != int(xyz_batch_cnt.sum()) or new_xyz.shape[0] != int(new_xyz_batch_cnt.sum()):
            raise AssertionError(f"rows and batch counts disagree: xyz {tuple(xyz.shape)} / {xyz_batch_cnt.tolist()}, "
                                 f"new_xyz {tuple(new_xyz.shape)} / {new_xyz_batch_cnt.tolist()}")
        idx, empty = ball_query(self.radius, self.nsample, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt)
        rel = grouping_operation(xyz, xyz_batch_cnt, idx, new_xyz_batch_cnt) - new_xyz.unsqueeze(-1)
        rel[empty] = 0
        if features is None:
            assert self.use_xyz, "Cannot have not features and not use xyz as a feature!"
            return rel, idx
        grouped = grouping_operation(features, xyz_batch_cnt, idx, new_xyz_batch_cnt)
        grouped[empty] = 0
        return (torch.cat([rel, grouped], dim=1) if self.use_xyz else grouped), idx
