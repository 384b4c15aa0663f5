"""Fused grid set-abstraction (csrc/sa_fused.hip): what PointnetSAModuleMSG.forward (pointnet2_modules.py:30-62) does after
its ball query, for the bn=False shared MLP of the RoI head, without the grouped (R, C, M, S) tensor:

    sa_grid_max(per_point, per_centre, idx, w2)[r, i, :] = max_s relu(w2 @ relu(per_point[r, idx[r, i, s]] - per_centre[r, i]))

per_point (R, N, 64) / per_centre (R, M, 64) are the first (linear) shared-MLP layer applied to the points and to the centres,
idx (R, M, S) the ball-query result, w2 (64, 64) the second layer's 1x1-conv weight.  Gradients for per_point, per_centre and
w2 (the forward pass keeps one byte per centre and channel: which sample attained the maximum).  GPU only; `supported()` tells whether the kernel covers a shape (64 channels, 16 or 32 samples, N <= 884; backward runs eight waves per workgroup up to N = 589)."""
import torch

import fv2p_native as _nat

from ... import _glue as G


def supported(per_point, idx):
    if not (per_point.is_cuda and per_point.dtype == torch.float32 and per_point.dim() == 3 and idx.dim() == 3):
        return False
    return bool(_nat.lib().fv2p_sa_grid_supported(per_point.shape[1], idx.shape[1], idx.shape[2], per_point.shape[2]))


def _fwd(saved, per_point, per_centre, idx, w2):
    r, n, c = per_point.shape
    m, s = idx.shape[1], idx.shape[2]
    per_point, per_centre, w2 = per_point.contiguous(), per_centre.contiguous(), w2.contiguous()
    out = G.new(per_point, (r, m, c))
    need = per_point.requires_grad or per_centre.requires_grad or w2.requires_grad
    arg = G.new(per_point, (r, m, c), torch.uint8) if need else None   # which sample attained each maximum: all that backward needs
    G.run("fv2p_sa_grid_fwd", per_point, per_centre, idx, w2, r, n, m, s, c, out, arg)
    saved.update(t=(per_point, per_centre, idx, w2, arg), dims=(r, n, m, s, c))
    return out


def _bwd(saved, grad):
    per_point, per_centre, idx, w2, arg = saved["t"]
    if arg is None:
        raise RuntimeError("sa_grid_max: backward of a forward pass that ran without gradients enabled")
    r, n, m, s, c = saved["dims"]
    g_point, g_centre, g_w = torch.empty_like(per_point), torch.empty_like(per_centre), torch.empty_like(w2)
    ws = G.scratch("fv2p_sa_grid_bwd_ws_bytes", grad.device, r)
    G.run("fv2p_sa_grid_bwd", per_point, per_centre, idx, w2, arg, grad.contiguous(), r, n, m, s, c, g_point, g_centre, g_w, ws, ws.numel())
    return g_point, g_centre, None, g_w


SaGridMax = G.autograd_op("SaGridMax", _fwd, _bwd)
sa_grid_max = SaGridMax.apply
