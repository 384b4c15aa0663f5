"""TEST INFRASTRUCTURE ONLY (oracle/): numpy float32 restatement of the reference's bilinear BEV gather,
`bilinear_interpolate_torch` (pcdet/models/backbones_3d/pfe/bev_grid_pooling.py:11-45) and the coordinate arithmetic of
`BEVGridPooling.interpolate_from_bev_features` (:68-83).  Pinned: tests/golden/bev_interp_*.npz hold the outputs of the
reference function itself (oracle/gen_golden_bev.py compiles that one function out of the reference file in the build
container); tests/test_oracle_golden.py checks this restatement against them bit for bit."""
import numpy as np


def bilinear_interpolate(im, x, y):
    """im (H, W, C) float32, x (N), y (N) float32 -> (N, C) float32; corners clamped to the map, weights from the
    clamped corners, products summed in the order a, b, c, d — all in float32 like the reference's torch ops."""
    im = np.asarray(im)
    im = im if im.dtype == np.float64 else im.astype(np.float32, copy=False)   # float64 maps: the calibration run of tests/f64_calibration.py only
    x, y = np.asarray(x, np.float32), np.asarray(y, np.float32)
    h, w = im.shape[0], im.shape[1]
    x0 = np.floor(x).astype(np.int64)
    y0 = np.floor(y).astype(np.int64)
    x1, y1 = x0 + 1, y0 + 1
    x0, x1 = np.clip(x0, 0, w - 1), np.clip(x1, 0, w - 1)
    y0, y1 = np.clip(y0, 0, h - 1), np.clip(y1, 0, h - 1)
    ia, ib, ic, id_ = im[y0, x0], im[y1, x0], im[y0, x1], im[y1, x1]
    f = lambda v: v.astype(np.float32)
    wa = (f(x1) - x) * (f(y1) - y)
    wb = (f(x1) - x) * (y - f(y0))
    wc = (x - f(x0)) * (f(y1) - y)
    wd = (x - f(x0)) * (y - f(y0))
    return ((ia * wa[:, None] + ib * wb[:, None]) + ic * wc[:, None]) + id_ * wd[:, None]


def pixel_coordinates(keypoints, bev_stride, point_cloud_range, voxel_size, gpu_scalar_division=True):
    """x_idxs, y_idxs of bev_grid_pooling.py:69-72.  The reference evaluates `tensor / python_scalar` with torch on the
    GPU, whose kernel multiplies by the fp32 reciprocal of the scalar (ATen BinaryDivTrueKernel.cu, CPU-scalar fast
    path); torch's CPU kernel divides.  Both are restated; the GPU form is what a detector run produces."""
    kp = np.asarray(keypoints, np.float32)
    one = np.float32(1.0)
    div = (lambda a, b: a * (one / np.float32(b))) if gpu_scalar_division else (lambda a, b: a / np.float32(b))
    xs = div(kp[:, :, 0] - np.float32(point_cloud_range[0]), voxel_size[0])
    ys = div(kp[:, :, 1] - np.float32(point_cloud_range[1]), voxel_size[1])
    return div(xs, bev_stride), div(ys, bev_stride)


def interpolate_from_bev_features(keypoints, bev_features, bev_stride, point_cloud_range, voxel_size, gpu_scalar_division=True):
    """keypoints (B, N, 3), bev_features (B, C, H, W) -> (B, N, C)."""
    kp = np.asarray(keypoints, np.float32)
    xs, ys = pixel_coordinates(kp, bev_stride, point_cloud_range, voxel_size, gpu_scalar_division)
    return np.stack([bilinear_interpolate(np.ascontiguousarray(np.transpose(bev_features[k], (1, 2, 0))), xs[k], ys[k])
                     for k in range(kp.shape[0])])


def bilinear_interpolate_grad(im_shape, x, y, grad_out):
    """Gradient of the map (H, W, C) in float64: the transposed scatter of the same four weights."""
    h, w, c = im_shape
    x, y = np.asarray(x, np.float32), np.asarray(y, np.float32)
    x0 = np.floor(x).astype(np.int64)
    y0 = np.floor(y).astype(np.int64)
    x1, y1 = x0 + 1, y0 + 1
    x0, x1 = np.clip(x0, 0, w - 1), np.clip(x1, 0, w - 1)
    y0, y1 = np.clip(y0, 0, h - 1), np.clip(y1, 0, h - 1)
    f = lambda v: v.astype(np.float32)
    ws = [(f(x1) - x) * (f(y1) - y), (f(x1) - x) * (y - f(y0)), (x - f(x0)) * (f(y1) - y), (x - f(x0)) * (y - f(y0))]
    g = np.zeros((h, w, c), np.float64)
    for (yy, xx), wt in zip([(y0, x0), (y1, x0), (y0, x1), (y1, x1)], ws):
        np.add.at(g, (yy, xx), np.asarray(grad_out, np.float64) * wt.astype(np.float64)[:, None])
    return g
