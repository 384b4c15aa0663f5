"""ORACLE — test infrastructure only: runs the product's Python layers on the CPU with every C-ABI call answered by the oracle.

`with oracle_backend():` swaps the five functions through which `pcdet.ops` reaches libfv2p_ops.so
(`fv2p_native.call / lib / stream / workspace / require_cuda`) for stand-ins that take CPU tensors and fill the
caller-allocated outputs from oracle/ (the reference algorithms restated).  The L2 glue, the autograd Functions and the
harness model therefore execute unchanged on the host and give the reference-side answer for a whole op sequence;
the GPU run of the same code is compared against it (tests/test_fv2p_step_gpu.py) and bench.py's cpu_baseline leg times it.
Only symbols the FV2P step (the harness's, or the reference's own model classes over this package: tests/test_reference_overlay.py) reaches are answered; anything else raises.  Never imported by the product package."""
import contextlib

import numpy as np
import torch

import fv2p_native
import oracle
from oracle import bev_oracle


def _np(t):
    return t.detach().cpu().numpy()


def _fill(dst, arr):
    dst.copy_(torch.from_numpy(np.ascontiguousarray(arr)).to(dst.dtype).view_as(dst))


def _three_nn_stack(b, n, m, unknown, ucnt, known, kcnt, dist2, idx, stream):
    d2, i = oracle.three_nn_stack(_np(unknown), _np(ucnt), _np(known), _np(kcnt))
    _fill(dist2, d2), _fill(idx, i)


def _three_nn_stack_grid(b, n, m, unknown, ucnt, known, kcnt, cell, dist2, idx, ws, ws_bytes, stream):
    _three_nn_stack(b, n, m, unknown, ucnt, known, kcnt, dist2, idx, stream)      # the grid search is the same function, bit for bit


def _three_interp_stack(n, c, feats, idx, weight, out, stream):
    f, i, w = _np(feats), _np(idx).astype(np.int64), _np(weight)
    _fill(out, (w[:, 0:1] * f[i[:, 0]] + w[:, 1:2] * f[i[:, 1]]) + w[:, 2:3] * f[i[:, 2]])   # interpolate_gpu.cu (stack): left to right


def _three_interp_stack_grad(n, c, grad_out, idx, weight, grad_feats, stream):
    g, i, w = _np(grad_out).astype(np.float64), _np(idx).astype(np.int64), _np(weight).astype(np.float64)
    acc = np.zeros(tuple(grad_feats.shape), np.float64)
    for k in range(3):
        np.add.at(acc, i[:, k], g * w[:, k:k + 1])
    _fill(grad_feats, acc)


def _three_interp_stack_grad_gather(n, c, m, grad_out, idx, weight, grad_feats, ws, ws_bytes, stream):
    _three_interp_stack_grad(n, c, grad_out, idx, weight, grad_feats, stream)


def _fps(b, n, m, xyz, temp, idxs, ws, ws_bytes, stream):
    out, t = oracle.furthest_point_sample(_np(xyz).reshape(b, n, 3), m)
    _fill(idxs, out), _fill(temp, t)


def _points_in_boxes(boxes, pts, b, t, n, out, stream):
    _fill(out, oracle.points_in_boxes_gpu(_np(pts).reshape(b, n, 3), _np(boxes).reshape(b, t, 7)))


def _nms(boxes, n, thresh, normal, keep, cnt, ws, ws_bytes, stream):
    bx = _np(boxes).reshape(n, 7)
    k = oracle.nms(bx, -np.arange(n, dtype=np.float32), thresh, None, bool(normal))   # boxes arrive sorted: keep their order
    keep.zero_()
    keep[:len(k)] = torch.from_numpy(np.asarray(k, np.int64))
    cnt.fill_(len(k))


def _nms_batch(boxes, b, n, thresh, normal, max_keep, keep, keep_stride, cnt, ws, ws_bytes, stream):
    bx = _np(boxes).reshape(b, n, 7)
    keep.zero_()
    for s in range(b):
        k = oracle.nms(bx[s], -np.arange(n, dtype=np.float32), thresh, None, bool(normal))
        if max_keep > 0:
            k = k[:max_keep]      # model_nms_utils.py:20: selected[:NMS_POST_MAXSIZE]
        keep[s, :len(k)] = torch.from_numpy(np.asarray(k, np.int64))
        cnt[s] = len(k)


def _overlap_bev(a, na, bb, nb, out, stream):
    _fill(out, oracle.boxes_bev(_np(a).reshape(na, 7), _np(bb).reshape(nb, 7), "overlap"))


def _roipoint(xyz, boxes, feats, b, n, m, c, s, pooled, flag, stream):
    p, f = oracle.roipoint_pool3d(_np(xyz).reshape(b, n, 3), _np(feats).reshape(b, n, c), _np(boxes).reshape(b, m, 7), s)
    _fill(pooled, p), _fill(flag, f)


def _ball_query_batch(b, n, m, radius, nsample, new_xyz, xyz, idx, stream):
    _fill(idx, oracle.ball_query_batch(radius, nsample, _np(xyz).reshape(b, n, 3), _np(new_xyz).reshape(b, m, 3)))


def _group_batch(b, c, n, npoints, nsample, points, idx, out, stream):
    _fill(out, oracle.group_points_batch(_np(points).reshape(b, c, n), _np(idx).reshape(b, npoints, nsample)))


def _group_batch_grad(b, c, n, npoints, nsample, grad_out, idx, grad_points, stream):
    g = _np(grad_out).reshape(b, c, npoints * nsample).astype(np.float64)
    i = _np(idx).reshape(b, npoints * nsample).astype(np.int64)
    acc = np.zeros((b, c, n), np.float64)
    for s in range(b):
        np.add.at(acc[s].T, i[s], g[s].T)
    _fill(grad_points, acc)


def _bev_fwd(bev, b, c, h, w, channels_first, x, y, n, out, ws, ws_bytes, stream):
    m = _np(bev).reshape((b, c, h, w) if channels_first else (b, h, w, c))
    xs, ys = _np(x).reshape(b, n), _np(y).reshape(b, n)
    res = [bev_oracle.bilinear_interpolate(np.transpose(m[k], (1, 2, 0)) if channels_first else m[k], xs[k], ys[k]) for k in range(b)]
    _fill(out, np.stack(res))


def _bev_bwd(grad_out, b, c, h, w, channels_first, x, y, n, grad_bev, ws, ws_bytes, stream):
    g = _np(grad_out).reshape(b, n, c)
    xs, ys = _np(x).reshape(b, n), _np(y).reshape(b, n)
    res = [bev_oracle.bilinear_interpolate_grad((h, w, c), xs[k], ys[k], g[k]) for k in range(b)]
    res = np.stack(res)      # float64 sums, rounded once to the caller's dtype
    _fill(grad_bev, np.transpose(res, (0, 3, 1, 2)) if channels_first else res)


def _three_nn_batch(b, n, m, unknown, known, dist2, idx, stream):
    d2, i = oracle.three_nn_batch(_np(unknown).reshape(b, n, 3), _np(known).reshape(b, m, 3))
    _fill(dist2, d2), _fill(idx, i)


def _three_interp_batch(b, c, m, n, points, idx, weight, out, stream):
    _fill(out, oracle.three_interpolate_batch(_np(points).reshape(b, c, m), _np(idx).reshape(b, n, 3), _np(weight).reshape(b, n, 3)))


def _three_interp_batch_grad(b, c, n, m, grad_out, idx, weight, grad_points, stream):
    g, i, w = _np(grad_out).reshape(b, c, n).astype(np.float64), _np(idx).reshape(b, n, 3).astype(np.int64), _np(weight).reshape(b, n, 3).astype(np.float64)
    acc = np.zeros((b, c, m), np.float64)
    for s in range(b):
        for k in range(3):
            np.add.at(acc[s].T, i[s, :, k], (g[s] * w[s, :, k]).T)          # interpolate_gpu.cu:107-149: grad_points[idx] += grad_out * weight
    _fill(grad_points, acc)


def _dcn_forward(x_nhwc, wt_oc, bias, offset, mask, b, h, w, cin, cout, ho, wo, kh, kw, sh, sw, ph, pw, dh, dw, dg, y_nhwc, stream):
    from oracle import dcn_oracle
    x = x_nhwc.view(b, h, w, cin).permute(0, 3, 1, 2)
    weight = wt_oc.view(kh, kw, cout, cin).permute(2, 3, 0, 1)   # the forward entry takes [kh*kw][Cout][Cin]
    y = dcn_oracle.modulated_deform_conv(x, offset.view(b, dg * 2 * kh * kw, ho, wo), mask.view(b, dg * kh * kw, ho, wo), weight, bias,
                                         (sh, sw), (ph, pw), (dh, dw), dg)
    y_nhwc.copy_(y.permute(0, 2, 3, 1).reshape(b * ho * wo, cout))


def _dcn_backward(x_nhwc, wt, offset, mask, dy_nhwc, b, h, w, cin, cout, ho, wo, kh, kw, sh, sw, ph, pw, dh, dw, dg, dx_nhwc, doffset, dmask, dwt,
                  ws, ws_bytes, stream):
    """modulated_deform_conv_cuda.cu:127-280 through autograd of the oracle's forward (float64 inside, one rounding at the end)."""
    from oracle import dcn_oracle
    with torch.enable_grad():
        x = x_nhwc.view(b, h, w, cin).permute(0, 3, 1, 2).double().requires_grad_(True)
        weight = wt.view(kh, kw, cin, cout).permute(3, 2, 0, 1).double().requires_grad_(True)
        off = offset.view(b, dg * 2 * kh * kw, ho, wo).double().requires_grad_(True)
        msk = mask.view(b, dg * kh * kw, ho, wo).double().requires_grad_(True)
        y = dcn_oracle.modulated_deform_conv(x, off, msk, weight, None, (sh, sw), (ph, pw), (dh, dw), dg)
        dy = dy_nhwc.view(b, ho, wo, cout).permute(0, 3, 1, 2).double()
        gx, gw, go, gm = torch.autograd.grad(y, (x, weight, off, msk), dy)
    dx_nhwc.copy_(gx.permute(0, 2, 3, 1).reshape(dx_nhwc.shape))      # copy_ rounds once to the caller's dtype
    doffset.copy_(go.reshape(doffset.shape))
    dmask.copy_(gm.reshape(dmask.shape))
    dwt.copy_(gw.permute(2, 3, 1, 0).reshape(dwt.shape))


_TABLE = {
    "fv2p_dcn_forward": _dcn_forward,
    "fv2p_dcn_backward": _dcn_backward,
    "fv2p_three_nn_batch": _three_nn_batch,
    "fv2p_three_interpolate_batch": _three_interp_batch,
    "fv2p_three_interpolate_batch_grad": _three_interp_batch_grad,
    "fv2p_three_nn_stack": _three_nn_stack,
    "fv2p_three_nn_stack_grid": _three_nn_stack_grid,
    "fv2p_three_interpolate_stack": _three_interp_stack,
    "fv2p_three_interpolate_stack_grad": _three_interp_stack_grad,
    "fv2p_three_interpolate_stack_grad_gather": _three_interp_stack_grad_gather,
    "fv2p_furthest_point_sampling": _fps,
    "fv2p_points_in_boxes": _points_in_boxes,
    "fv2p_nms": _nms,
    "fv2p_nms_batch": _nms_batch,
    "fv2p_boxes_overlap_bev": _overlap_bev,
    "fv2p_roipoint_pool3d": _roipoint,
    "fv2p_ball_query_batch": _ball_query_batch,
    "fv2p_group_points_batch": _group_batch,
    "fv2p_group_points_batch_grad": _group_batch_grad,
    "fv2p_bev_interp_fwd": _bev_fwd,
    "fv2p_bev_interp_bwd": _bev_bwd,
}


def _call(name, *args):
    fn = _TABLE.get(name)
    if fn is None:
        raise NotImplementedError(f"oracle backend: {name} is not answered (add it to oracle/backend.py)")
    with torch.no_grad():
        fn(*args)
    return 0


class _FakeLib:
    def __getattr__(self, name):
        if name.endswith("_ws_bytes"):
            return lambda *a: 16
        raise AttributeError(name)


@contextlib.contextmanager
def oracle_backend():
    saved = {k: getattr(fv2p_native, k) for k in ("call", "lib", "stream", "workspace", "require_cuda", "device_guard")}
    fv2p_native.call = _call
    fv2p_native.lib = lambda: _FakeLib()
    fv2p_native.stream = lambda: 0
    fv2p_native.workspace = lambda nbytes, device: torch.empty(max(int(nbytes), 16), dtype=torch.uint8)
    fv2p_native.require_cuda = lambda *t: None
    fv2p_native.device_guard = lambda device: contextlib.nullcontext()
    try:
        yield
    finally:
        for k, v in saved.items():
            setattr(fv2p_native, k, v)
