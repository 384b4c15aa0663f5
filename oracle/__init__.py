"""ORACLE — test infrastructure only (see each source file's header).

CPU restatements of the reference algorithms used as the parity checker by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg.  Never imported by the product
package (from-voxel-to-point_amd/)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith(".c")]
        if not os.path.exists(path) or any(os.path.getmtime(s) > os.path.getmtime(path) for s in srcs):
            build()
        _LIB = ctypes.CDLL(path)
    return _LIB


import contextlib


@contextlib.contextmanager
def contracted():
    """Inside the block every oracle function runs from liboracle_fma.so: the same C with fused multiply-adds wherever a product
    feeds a sum (gcc -ffp-contract=fast -mfma), the arithmetic an nvcc build of the reference's kernels performs by default.
    For the contraction audit only (tools/fma_audit.py)."""
    global _LIB
    path = os.path.join(_HERE, "liboracle_fma.so")
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith(".c")]
    if not os.path.exists(path) or any(os.path.getmtime(s) > os.path.getmtime(path) for s in srcs):
        subprocess.check_call(["make", "-s", "-C", _HERE, "fma"])
    plain = lib()
    _LIB = ctypes.CDLL(path)
    try:
        yield
    finally:
        _LIB = plain


def _p(a, t):
    return a.ctypes.data_as(ctypes.POINTER(t))


_f32, _i32, _i64 = ctypes.c_float, ctypes.c_int32, ctypes.c_int64


def points_to_voxel(points, voxel_size, coors_range, max_points, max_voxels, scratch_map=None):
    """voxel_generator.py:75-133 restated (reverse_index=True). Returns (voxels, coors, num_points)."""
    points = np.ascontiguousarray(points, dtype=np.float32)
    vs = np.ascontiguousarray(voxel_size, dtype=np.float32)
    rng = np.ascontiguousarray(coors_range, dtype=np.float32)
    n, ndim = points.shape
    voxels = np.zeros((max_voxels, max_points, ndim), np.float32)
    coors = np.zeros((max_voxels, 3), np.int32)
    num = np.zeros((max_voxels,), np.int32)
    f = lib().oracle_points_to_voxel
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.POINTER(_f32), _i64, ctypes.c_int, ctypes.POINTER(_f32), ctypes.POINTER(_f32), ctypes.c_int,
                  ctypes.c_int, ctypes.POINTER(_f32), ctypes.POINTER(_i32), ctypes.POINTER(_i32), ctypes.POINTER(_i32)]
    sm = _p(scratch_map, _i32) if scratch_map is not None else None
    m = f(_p(points, _f32), n, ndim, _p(vs, _f32), _p(rng, _f32), max_points, max_voxels, _p(voxels, _f32),
          _p(coors, _i32), _p(num, _i32), sm)
    if m < 0:
        raise MemoryError("oracle_points_to_voxel")
    return voxels[:m], coors[:m], num[:m]


# ---------------------------------------------------------------- spconv rulebook -----------
def _conv_out_shape(shape, k, s, p, d):
    return [(shape[i] + 2 * p[i] - d[i] * (k[i] - 1) - 1) // s[i] + 1 for i in range(len(shape))]


def _deconv_out_shape(shape, k, s, p, d, op):
    return [(shape[i] - 1) * s[i] - 2 * p[i] + k[i] + op[i] for i in range(len(shape))]


def indice_pairs(indices, batch_size, spatial_shape, ksize, stride, padding, dilation, out_padding=(0, 0, 0),
                 subm=False, transpose=False, canonical=True, force_sparse=False):
    """spconv_ops.h:27-140 (getIndicePair<3>) on the CPU functors of geometry.h, restated in rulebook_oracle.c.
    canonical=True returns the GPU reference's output order (sorted flat index, spconv_ops.h:130) for conv;
    canonical=False keeps the CPU reference's first-touch order.  Returns (outids [M,4], pairs [K,2,N], num [K])."""
    ind = np.ascontiguousarray(indices, dtype=np.int32)
    n = ind.shape[0]
    i3 = lambda v: np.ascontiguousarray(v, dtype=np.int32)
    k, d = i3(ksize), i3(dilation)
    kvol = int(np.prod(k))
    pairs = np.full((kvol, 2, max(n, 1)), -1, np.int32)[:, :, :n].copy() if n else np.full((kvol, 2, 0), -1, np.int32)
    pairs = np.ascontiguousarray(pairs)
    num = np.zeros((kvol,), np.int32)
    L = lib()
    pi = ctypes.POINTER(_i32)
    if subm:
        s, p = i3([1, 1, 1]), i3([kk // 2 for kk in ksize])  # spconv_ops.h:76-80
        out_shape = i3(spatial_shape)
        f = L.oracle_indice_pairs_subm
        f.restype = ctypes.c_int
        f.argtypes = [pi, _i32, _i32, pi, pi, pi, pi, pi, pi, pi, ctypes.c_int]
        r = f(_p(ind, _i32), n, batch_size, _p(k, _i32), _p(s, _i32), _p(p, _i32), _p(d, _i32), _p(out_shape, _i32),
              _p(pairs, _i32), _p(num, _i32), int(force_sparse))
        assert r >= 0
        return ind.copy(), pairs, num
    s, p = i3(stride), i3(padding)
    if transpose:
        out_shape = i3(_deconv_out_shape(list(spatial_shape), list(ksize), list(stride), list(padding), list(dilation), list(out_padding)))
    else:
        out_shape = i3(_conv_out_shape(list(spatial_shape), list(ksize), list(stride), list(padding), list(dilation)))
    out_ids = np.zeros((max(n * kvol, 1), 4), np.int32)
    f = L.oracle_indice_pairs_conv
    f.restype = ctypes.c_int
    f.argtypes = [pi, _i32, _i32, pi, pi, pi, pi, pi, ctypes.c_int, pi, pi, pi, ctypes.c_int]
    m = f(_p(ind, _i32), n, batch_size, _p(k, _i32), _p(s, _i32), _p(p, _i32), _p(d, _i32), _p(out_shape, _i32),
          int(transpose), _p(out_ids, _i32), _p(pairs, _i32), _p(num, _i32), int(force_sparse))
    assert m >= 0
    if canonical:
        g = L.oracle_canonicalize_conv
        g.restype = ctypes.c_int
        g.argtypes = [pi, _i32, pi, pi, pi, _i32, _i32]
        assert g(_p(out_ids, _i32), m, _p(out_shape, _i32), _p(pairs, _i32), _p(num, _i32), kvol, n) == 0
    return out_ids[:m].copy(), pairs, num


def _work_dtype(x):
    """float32 — the reference's arithmetic — unless the caller hands float64 data: the float64 pass exists only to CALIBRATE the
    step-level parity tests (tests/f64_calibration.py: how far is a float32 run of this very network from its float64 run), it is
    never the thing compared bit for bit and never a reference restatement."""
    import torch
    if isinstance(x, torch.Tensor):
        return torch.float64 if x.dtype == torch.float64 else torch.float32
    return torch.float64 if np.asarray(x).dtype == np.float64 else torch.float32


def _np_work_dtype(x):
    return np.float64 if np.asarray(x).dtype == np.float64 else np.float32


def indice_conv(features, filters, pairs, num, n_out, inverse=False, subm=False):
    """spconv_ops.h:260-362 (indiceConv<float>) restated with torch CPU ops: zeros output, subM centre GEMM
    first (:300-304), then k = 0..K-1 gather -> mm -> scatter-add (:308-357).  fp32."""
    import torch
    dt = _work_dtype(features)
    feats = torch.as_tensor(features, dtype=dt)
    w = torch.as_tensor(filters, dtype=dt)
    cin, cout = w.shape[-2], w.shape[-1]
    w = w.reshape(-1, cin, cout)
    pairs_t = torch.as_tensor(np.asarray(pairs)).long()
    num = np.asarray(num)
    kvol = w.shape[0]
    out = torch.zeros((n_out, cout), dtype=dt)
    centre = int(np.argmax(num)) if subm else -1  # spconv_ops.h:272-277
    if subm:
        out = torch.mm(feats, w[centre])
    a, b = (1, 0) if inverse else (0, 1)
    for k in range(kvol):
        nh = int(num[k])
        if nh <= 0 or (subm and k == centre):
            continue
        out.index_add_(0, pairs_t[k, b, :nh], torch.mm(feats[pairs_t[k, a, :nh]], w[k]))
    return out


def indice_conv_backward(features, filters, out_bp, pairs, num, inverse=False, subm=False):
    """spconv_ops.h:364-457 restated: dW_k = gather(feat)^T gather(dout); dX[in] += dout[out] W_k^T."""
    import torch
    dt = _work_dtype(features)
    feats = torch.as_tensor(features, dtype=dt)
    w = torch.as_tensor(filters, dtype=dt)
    g = torch.as_tensor(out_bp, dtype=dt)
    cin, cout = w.shape[-2], w.shape[-1]
    w3 = w.reshape(-1, cin, cout)
    pairs_t = torch.as_tensor(np.asarray(pairs)).long()
    num = np.asarray(num)
    din = torch.zeros_like(feats)
    dw = torch.zeros_like(w3)
    a, b = (1, 0) if inverse else (0, 1)
    for k in range(w3.shape[0]):
        nh = int(num[k])
        if nh <= 0:
            continue
        fi, go = feats[pairs_t[k, a, :nh]], g[pairs_t[k, b, :nh]]
        dw[k] = fi.t() @ go
        din.index_add_(0, pairs_t[k, a, :nh], go @ w3[k].t())
    return din, dw.reshape(w.shape)


def indice_maxpool(features, pairs, num, n_out):
    """pool_ops.h:25-57 + maxpool.cc: zero-initialised output, out = max(out, in) per pair."""
    f = np.asarray(features, dtype=np.float32)
    out = np.zeros((n_out, f.shape[1]), np.float32)
    for k in range(pairs.shape[0]):
        nh = int(num[k])
        if nh:
            np.maximum.at(out, pairs[k, 1, :nh], f[pairs[k, 0, :nh]])
    return out


def indice_maxpool_backward(features, out_features, out_bp, pairs, num):
    """pool_ops.h:59-94: din[i] += dout[o] where in[i] == out[o]."""
    f, o, g = (np.asarray(x, dtype=np.float32) for x in (features, out_features, out_bp))
    din = np.zeros_like(f)
    for k in range(pairs.shape[0]):
        nh = int(num[k])
        if nh:
            i, oo = pairs[k, 0, :nh], pairs[k, 1, :nh]
            np.add.at(din, i, np.where(f[i] == o[oo], g[oo], 0).astype(np.float32))
    return din


def indice_group(features, pairs, num, n_out):
    """group_ops.h:29-140: out[k, o, :] = feat[i, :] for every pair (i, o) of offset k, zeros elsewhere."""
    f = np.asarray(features, dtype=np.float32)
    out = np.zeros((pairs.shape[0], n_out, f.shape[1]), np.float32)
    for k in range(pairs.shape[0]):
        nh = int(num[k])
        if nh:
            out[k, pairs[k, 1, :nh]] = f[pairs[k, 0, :nh]]
    return out


# ---------------------------------------------------------------- iou3d / NMS ---------------
def boxes_bev(boxes_a, boxes_b, mode="iou", threads=None):
    """iou3d_nms_kernel.cu:236-265 restated: pairwise BEV overlap area (mode='overlap') or IoU (mode='iou').
    threads: rows in parallel with OpenMP (bench.py's all-core B2 baseline); None = the serial form."""
    a = np.ascontiguousarray(boxes_a, dtype=np.float32)
    b = np.ascontiguousarray(boxes_b, dtype=np.float32)
    out = np.zeros((a.shape[0], b.shape[0]), np.float32)
    if threads:
        f = lib().oracle_boxes_bev_mt
        f.restype = None
        f.argtypes = [ctypes.POINTER(_f32), ctypes.c_int, ctypes.POINTER(_f32), ctypes.c_int, ctypes.c_int, ctypes.POINTER(_f32), ctypes.c_int]
        f(_p(a, _f32), a.shape[0], _p(b, _f32), b.shape[0], 1 if mode == "iou" else 0, _p(out, _f32), int(threads))
        return out
    f = lib().oracle_boxes_bev
    f.restype = None
    f.argtypes = [ctypes.POINTER(_f32), ctypes.c_int, ctypes.POINTER(_f32), ctypes.c_int, ctypes.c_int, ctypes.POINTER(_f32)]
    f(_p(a, _f32), a.shape[0], _p(b, _f32), b.shape[0], 1 if mode == "iou" else 0, _p(out, _f32))
    return out


def boxes_iou3d(boxes_a, boxes_b):
    """iou3d_nms_utils.py:454-491 restated in numpy float32."""
    a = np.asarray(boxes_a, np.float32)
    b = np.asarray(boxes_b, np.float32)
    ov = boxes_bev(a, b, "overlap")
    a_max, a_min = (a[:, 2] + a[:, 5] / 2)[:, None], (a[:, 2] - a[:, 5] / 2)[:, None]
    b_max, b_min = (b[:, 2] + b[:, 5] / 2)[None, :], (b[:, 2] - b[:, 5] / 2)[None, :]
    oh = np.clip(np.minimum(a_max, b_max) - np.maximum(a_min, b_min), 0, None)
    o3 = ov * oh
    va, vb = (a[:, 3] * a[:, 4] * a[:, 5])[:, None], (b[:, 3] * b[:, 4] * b[:, 5])[None, :]
    return np.clip(o3 / np.clip(va + vb - o3, 1e-6, None), 0, 1).astype(np.float32)


def nms(boxes, scores, thresh, pre_maxsize=None, normal=False, threads=None):
    """iou3d_nms_utils.py:494-526 restated: stable descending score sort, optional cut, greedy suppression.
    Returns indices into `boxes`.  threads: the reference's two phases with the pair mask computed by OpenMP rows
    (bench.py's all-core B3 baseline; same survivors); None = the serial form that evaluates only the pairs it reaches."""
    boxes = np.ascontiguousarray(boxes, dtype=np.float32)
    order = np.argsort(-np.asarray(scores, np.float32), kind="stable")
    if pre_maxsize is not None:
        order = order[:pre_maxsize]
    sb = np.ascontiguousarray(boxes[order])
    keep = np.zeros((max(sb.shape[0], 1),), np.int64)
    if threads:
        f = lib().oracle_nms_mt
        f.restype = ctypes.c_int
        f.argtypes = [ctypes.POINTER(_f32), ctypes.c_int, ctypes.c_float, ctypes.c_int, ctypes.POINTER(_i64), ctypes.c_int]
        num = f(_p(sb, _f32), sb.shape[0], float(thresh), int(normal), _p(keep, _i64), int(threads))
        return order[keep[:num]]
    f = lib().oracle_nms
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.POINTER(_f32), ctypes.c_int, ctypes.c_float, ctypes.c_int, ctypes.POINTER(_i64)]
    num = f(_p(sb, _f32), sb.shape[0], float(thresh), int(normal), _p(keep, _i64))
    return order[keep[:num]]


# ---------------------------------------------------------------- roiaware / roipoint ---------
def points_in_boxes_gpu(points, boxes):
    """roiaware_pool3d_kernel.cu:313-336 restated: points (B,M,3), boxes (B,T,7) -> (B,M) first box index or -1."""
    p = np.ascontiguousarray(points, np.float32)
    b = np.ascontiguousarray(boxes, np.float32)
    out = np.full((p.shape[0], p.shape[1]), -1, np.int32)
    f = lib().oracle_points_in_boxes_gpu
    f.restype = None
    f.argtypes = [ctypes.POINTER(_f32), ctypes.POINTER(_f32), ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(_i32)]
    f(_p(b, _f32), _p(p, _f32), p.shape[0], b.shape[1], p.shape[1], _p(out, _i32))
    return out


def points_in_boxes_cpu(points, boxes):
    """roiaware_pool3d.cpp:143-168 restated: points (M,3), boxes (N,7) -> (N,M) 0/1 with MARGIN 1e-2."""
    p = np.ascontiguousarray(points, np.float32)
    b = np.ascontiguousarray(boxes, np.float32)
    out = np.zeros((b.shape[0], p.shape[0]), np.int32)
    f = lib().oracle_points_in_boxes_cpu
    f.restype = None
    f.argtypes = [ctypes.POINTER(_f32), ctypes.POINTER(_f32), ctypes.c_int, ctypes.c_int, ctypes.POINTER(_i32)]
    f(_p(b, _f32), _p(p, _f32), b.shape[0], p.shape[0], _p(out, _i32))
    return out


def roipoint_pool3d(points, point_features, pooled_boxes3d, num_sampled_points=512):
    """roipoint_pool3d_kernel.cu:38-130 restated on already-enlarged boxes."""
    xyz = np.ascontiguousarray(points, np.float32)
    ft = np.ascontiguousarray(point_features, np.float32)
    bx = np.ascontiguousarray(pooled_boxes3d, np.float32)
    B, N, _ = xyz.shape
    M, C = bx.shape[1], ft.shape[2]
    pooled = np.zeros((B, M, num_sampled_points, 3 + C), np.float32)
    flag = np.zeros((B, M), np.int32)
    f = lib().oracle_roipoint_pool3d
    f.restype = None
    f.argtypes = [ctypes.POINTER(_f32)] * 3 + [ctypes.c_int] * 5 + [ctypes.POINTER(_f32), ctypes.POINTER(_i32)]
    f(_p(xyz, _f32), _p(bx, _f32), _p(ft, _f32), B, N, M, C, num_sampled_points, _p(pooled, _f32), _p(flag, _i32))
    return pooled, flag


def roiaware_pool3d(rois, pts, pts_feature, out_size, max_pts_each_voxel, pool_method):
    """roiaware_pool3d_kernel.cu:39-232 restated. Returns (pooled, argmax, pts_idx_of_voxels)."""
    r = np.ascontiguousarray(rois, np.float32)
    p = np.ascontiguousarray(pts, np.float32)
    ft = np.ascontiguousarray(pts_feature, np.float32)
    ox, oy, oz = (out_size,) * 3 if isinstance(out_size, int) else out_size
    R, C = r.shape[0], ft.shape[1]
    pooled = np.zeros((R, ox, oy, oz, C), np.float32)
    argmax = np.zeros((R, ox, oy, oz, C), np.int32)
    vox = np.zeros((R, ox, oy, oz, max_pts_each_voxel), np.int32)
    f = lib().oracle_roiaware_pool3d
    f.restype = None
    f.argtypes = [ctypes.POINTER(_f32)] * 3 + [ctypes.c_int] * 8 + [ctypes.POINTER(_i32), ctypes.POINTER(_i32), ctypes.POINTER(_f32)]
    f(_p(r, _f32), _p(p, _f32), _p(ft, _f32), R, p.shape[0], C, max_pts_each_voxel, ox, oy, oz, {"max": 0, "avg": 1}[pool_method],
      _p(argmax, _i32), _p(vox, _i32), _p(pooled, _f32))
    return pooled, argmax, vox


# ---------------------------------------------------------------- pointnet2 -------------------
def _pf(a):
    return np.ascontiguousarray(a, np.float32)


def _pi(a):
    return np.ascontiguousarray(a, np.int32)


_PF, _PI = ctypes.POINTER(_f32), ctypes.POINTER(_i32)


def ball_query_batch(radius, nsample, xyz, new_xyz):
    xyz, new_xyz = _pf(xyz), _pf(new_xyz)
    B, N, _ = xyz.shape
    M = new_xyz.shape[1]
    idx = np.zeros((B, M, nsample), np.int32)
    f = lib().oracle_ball_query_batch
    f.restype = None
    f.argtypes = [ctypes.c_int] * 3 + [ctypes.c_float, ctypes.c_int, _PF, _PF, _PI]
    f(B, N, M, radius, nsample, _p(new_xyz, _f32), _p(xyz, _f32), _p(idx, _i32))
    return idx


def ball_query_stack(radius, nsample, xyz, xyz_cnt, new_xyz, new_cnt):
    """Raw kernel output (idx[:,0] == -1 marks an empty ball), before pointnet2_utils.py:35-38 zeroes it."""
    xyz, new_xyz, xyz_cnt, new_cnt = _pf(xyz), _pf(new_xyz), _pi(xyz_cnt), _pi(new_cnt)
    M = new_xyz.shape[0]
    idx = np.zeros((M, nsample), np.int32)
    f = lib().oracle_ball_query_stack
    f.restype = None
    f.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_int, _PF, _PI, _PF, _PI, _PI]
    f(len(xyz_cnt), M, radius, nsample, _p(new_xyz, _f32), _p(new_cnt, _i32), _p(xyz, _f32), _p(xyz_cnt, _i32), _p(idx, _i32))
    return idx


def voxel_query_stack(max_range, radius, nsample, xyz, new_xyz, new_coords, point_indices):
    xyz, new_xyz, new_coords, point_indices = _pf(xyz), _pf(new_xyz), _pi(new_coords), _pi(point_indices)
    M = new_coords.shape[0]
    B, Z, Y, X = point_indices.shape
    idx = np.zeros((M, nsample), np.int32)
    f = lib().oracle_voxel_query_stack
    f.restype = None
    f.argtypes = [ctypes.c_int] * 5 + [ctypes.c_float] + [ctypes.c_int] * 3 + [_PF, _PF, _PI, _PI, _PI]
    f(M, Z, Y, X, nsample, radius, max_range[0], max_range[1], max_range[2], _p(new_xyz, _f32), _p(xyz, _f32), _p(new_coords, _i32),
      _p(point_indices, _i32), _p(idx, _i32))
    return idx


def furthest_point_sample(xyz, npoint):
    xyz = _pf(xyz)
    B, N, _ = xyz.shape
    temp = np.full((B, N), 1e10, np.float32)
    out = np.zeros((B, npoint), np.int32)
    f = lib().oracle_furthest_point_sampling
    f.restype = None
    f.argtypes = [ctypes.c_int] * 3 + [_PF, _PF, _PI]
    f(B, N, npoint, _p(xyz, _f32), _p(temp, _f32), _p(out, _i32))
    return out, temp


def three_nn_batch(unknown, known):
    """Returns (dist2, idx) — squared distances as the kernel writes them (the python layer takes sqrt)."""
    unknown, known = _pf(unknown), _pf(known)
    B, N, _ = unknown.shape
    d2 = np.zeros((B, N, 3), np.float32)
    idx = np.zeros((B, N, 3), np.int32)
    f = lib().oracle_three_nn_batch
    f.restype = None
    f.argtypes = [ctypes.c_int] * 3 + [_PF, _PF, _PF, _PI]
    f(B, N, known.shape[1], _p(unknown, _f32), _p(known, _f32), _p(d2, _f32), _p(idx, _i32))
    return d2, idx


def three_nn_stack(unknown, unk_cnt, known, known_cnt):
    unknown, known, unk_cnt, known_cnt = _pf(unknown), _pf(known), _pi(unk_cnt), _pi(known_cnt)
    N = unknown.shape[0]
    d2 = np.zeros((N, 3), np.float32)
    idx = np.zeros((N, 3), np.int32)
    f = lib().oracle_three_nn_stack
    f.restype = None
    f.argtypes = [ctypes.c_int, ctypes.c_int, _PF, _PI, _PF, _PI, _PF, _PI]
    f(len(unk_cnt), N, _p(unknown, _f32), _p(unk_cnt, _i32), _p(known, _f32), _p(known_cnt, _i32), _p(d2, _f32), _p(idx, _i32))
    return d2, idx


def three_interpolate_batch(features, idx, weight):
    """interpolate_gpu.cu:84-104: features (B,C,M), idx/weight (B,N,3) -> (B,C,N), fp32 left-to-right sum."""
    dt = _np_work_dtype(features)
    f, w = np.asarray(features, dt), np.asarray(weight, dt)
    idx = np.asarray(idx)
    g = np.take_along_axis(f[:, :, None, :], idx[:, None, :, :].astype(np.int64), axis=3)  # (B,C,N,3)
    return ((w[:, None, :, 0] * g[..., 0] + w[:, None, :, 1] * g[..., 1]) + w[:, None, :, 2] * g[..., 2]).astype(dt)


def group_points_batch(features, idx):
    """group_points_gpu.cu:53-72: features (B,C,N), idx (B,npoint,nsample) -> (B,C,npoint,nsample)."""
    f = np.asarray(features, _np_work_dtype(features))
    idx = np.asarray(idx).astype(np.int64)
    B, C, N = f.shape
    return np.stack([f[b][:, idx[b]] for b in range(B)], 0)
