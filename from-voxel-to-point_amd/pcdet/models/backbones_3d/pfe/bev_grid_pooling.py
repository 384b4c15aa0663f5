"""Bilinear gather of BEV features at key points on the GPU.

Mirrors the two callables of the reference's pcdet/models/backbones_3d/pfe/bev_grid_pooling.py that do the work:
``bilinear_interpolate_torch(im, x, y)`` (:11-45) and ``BEVGridPooling.interpolate_from_bev_features`` (:68-83, here also
as the free function ``interpolate_from_bev_features``).  Same arguments, same arithmetic (corners clamped to the map,
weights from the clamped corners), kernels in csrc/bev_interp.hip: no per-sample permute copy of the map, no four
[N, C] corner temporaries, one launch for the whole batch.  CUDA float32 only; anything else raises (no CPU fallback)."""
import torch
from torch import nn
from torch.autograd import Function

import fv2p_native as _nat


class _BevInterp(Function):
    """bev [B, C, H, W] or [B, H, W, C], x / y [B, N] pixel coordinates -> [B, N, C]; gradient for bev only."""

    @staticmethod
    def forward(ctx, bev, x, y, channels_first):
        _nat.require_cuda(bev, x, y)
        if bev.is_cuda and bev.dtype != torch.float32:
            raise _nat.Fv2pError("bev_grid_pooling: float32 feature maps expected")
        bev = bev.contiguous()
        x, y = x.detach().float().contiguous(), y.detach().float().contiguous()
        if channels_first:
            b, c, h, w = bev.shape
        else:
            b, h, w, c = bev.shape
        n = x.shape[1]
        out = torch.empty((b, n, c), dtype=bev.dtype, device=bev.device)
        with _nat.device_guard(bev.device):
            ws = _nat.workspace(max(int(_nat.lib().fv2p_bev_interp_ws_bytes(b, c, h, w, int(channels_first))), 16), bev.device)
            _nat.call("fv2p_bev_interp_fwd", bev, b, c, h, w, int(channels_first), x, y, n, out, ws, ws.numel(), _nat.stream())
        ctx.save_for_backward(x, y)
        ctx.geom = (b, c, h, w, bool(channels_first))
        return out

    @staticmethod
    def backward(ctx, grad_out):
        x, y = ctx.saved_tensors
        b, c, h, w, channels_first = ctx.geom
        g = grad_out.contiguous()
        grad_bev = torch.empty((b, c, h, w) if channels_first else (b, h, w, c), dtype=g.dtype, device=g.device)
        with _nat.device_guard(g.device):
            ws = _nat.workspace(max(b * c * h * w * 4 if channels_first else 16, 16), g.device)
            _nat.call("fv2p_bev_interp_bwd", g, b, c, h, w, int(channels_first), x, y, x.shape[1], grad_bev, ws, ws.numel(), _nat.stream())
        return grad_bev, None, None, None


def bilinear_interpolate_torch(im, x, y):
    """im (H, W, C), x (N), y (N) -> (N, C): the reference signature (bev_grid_pooling.py:11-45)."""
    return _BevInterp.apply(im.unsqueeze(0), x.reshape(1, -1), y.reshape(1, -1), False)[0]


def interpolate_from_bev_features(keypoints, bev_features, batch_size, bev_stride, point_cloud_range, voxel_size):
    """keypoints (B, N, 3), bev_features (B, C, H, W) -> (B, N, C); coordinate arithmetic as bev_grid_pooling.py:69-72
    (subtract the range origin, divide by the voxel size, then by the stride: three separate fp32 operations)."""
    x_idxs = (keypoints[:, :, 0] - point_cloud_range[0]) / voxel_size[0]
    y_idxs = (keypoints[:, :, 1] - point_cloud_range[1]) / voxel_size[1]
    x_idxs = x_idxs / bev_stride
    y_idxs = y_idxs / bev_stride
    assert bev_features.shape[0] == batch_size == keypoints.shape[0]
    return _BevInterp.apply(bev_features, x_idxs, y_idxs, True)


class BEVGridPooling(nn.Module):
    """The reference module's constructor arguments and outputs (bev_grid_pooling.py:48-125): optional
    Linear + BatchNorm1d + ReLU compression of the gathered features."""

    def __init__(self, model_cfg, point_cloud_range, voxel_size, **kwargs):
        super().__init__()
        self.model_cfg, self.point_cloud_range, self.voxel_size = model_cfg, point_cloud_range, voxel_size
        c_in, c_out = self.model_cfg.IN_CHANNELS, self.model_cfg.OUT_CHANNELS
        layers = [] if c_in == c_out else [nn.Linear(c_in, c_out, bias=False), nn.BatchNorm1d(c_out, eps=1e-3, momentum=0.01), nn.ReLU()]
        self.point_bev_feature_compress = nn.Sequential(*layers)
        self.num_point_bev_features = c_out

    def interpolate_from_bev_features(self, keypoints, bev_features, batch_size, bev_stride):
        return interpolate_from_bev_features(keypoints, bev_features, batch_size, bev_stride, self.point_cloud_range, self.voxel_size)

    def forward(self, batch_dict, keypoints):
        batch_size, num_keypoints, _ = keypoints.shape
        feats = self.interpolate_from_bev_features(keypoints, batch_dict['spatial_features_before_head'], batch_size,
                                                   bev_stride=batch_dict['spatial_features_stride'])
        feats = feats.view(batch_size * num_keypoints, -1)
        mods = list(self.point_bev_feature_compress)
        if len(mods) == 3:   # Linear -> BatchNorm1d -> ReLU: the pair after the GEMM as one fused op where it applies
            from pcdet.ops.spconv.norm import batch_norm_relu
            feats = mods[0](feats)
            fused = batch_norm_relu(mods[1], feats, mods[2])
            feats = fused if fused is not None else mods[2](mods[1](feats))
        return feats.view(batch_size, num_keypoints, -1)
