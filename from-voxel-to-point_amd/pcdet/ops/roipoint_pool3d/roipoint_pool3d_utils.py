"""RoIPointPool3d behind the names of the reference's pcdet/ops/roipoint_pool3d/roipoint_pool3d_utils.py:9-66 (the point
stream of FV2P's IoUGuidedRoIHead, iouguided_roi_head.py:28,179), on the C ABI (pcdet/ops/_glue.py)."""
import torch
import torch.nn as nn

from .. import _glue as G


def _enlarged(boxes3d, extra):
    """A list widens every side by a fixed amount (box_utils.enlarge_box3d); a scalar scales the sides (expand_box3d)."""
    big = boxes3d.clone()
    if isinstance(extra, (list, tuple)):
        # Python scalars become kernel arguments; a tensor built from the list would be a blocking host-to-device copy
        if len(set(extra)) == 1:
            big[..., 3:6] += extra[0]
        else:
            for axis, e in enumerate(extra):
                big[..., 3 + axis] += e
    else:
        big[..., 3:6] += boxes3d[..., 3:6] * extra
    return big


def _pool(saved, points, point_features, boxes3d, pool_extra_width, num_sampled_points=512):
    """points (B, N, 3), point_features (B, N, C), boxes3d (B, M, 7) -> (pooled (B, M, S, 3 + C), empty flag (B, M) int32):
    per enlarged box the first S inside points in index order, wrapped around when there are fewer."""
    if points.dim() != 3 or points.shape[2] != 3:
        raise AssertionError("RoIPointPool3d: points must be (B, N, 3)")
    b, n, _ = points.shape
    m, c = boxes3d.shape[1], point_features.shape[2]
    # the kernel writes every element (zeros for an empty box): no 100 MB fill in front of it, as the reference needs (:54)
    alloc = torch.empty if point_features.is_cuda else torch.zeros
    pooled = alloc((b, m, num_sampled_points, 3 + c), dtype=point_features.dtype, device=point_features.device)
    empty = torch.zeros((b, m), dtype=torch.int32, device=point_features.device)
    G.run("fv2p_roipoint_pool3d", points.contiguous(), _enlarged(boxes3d, pool_extra_width).contiguous(), point_features.contiguous(),
          b, n, m, c, num_sampled_points, pooled, empty)
    return pooled, empty


def _no_grad(saved, *grads):
    raise NotImplementedError   # as the reference (:65-66): the pool is used under no_grad


RoIPointPool3dFunction = G.autograd_op("RoIPointPool3dFunction", _pool, _no_grad)


class RoIPointPool3d(nn.Module):
    def __init__(self, num_sampled_points=512, pool_extra_width=1.0):
        super().__init__()
        self.num_sampled_points, self.pool_extra_width = num_sampled_points, pool_extra_width

    def forward(self, points, point_features, boxes3d):
        return RoIPointPool3dFunction.apply(points, point_features, boxes3d, self.pool_extra_width, self.num_sampled_points)
