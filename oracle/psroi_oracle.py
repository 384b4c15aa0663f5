"""ORACLE — test infrastructure only.

Deformable position-sensitive RoI pooling restated in numpy, every bin of the output at once and the samples of a bin in
the reference's order, in the dtype of the input (float32 = the reference's arithmetic operation for operation;
float64 for finite differences):
  geometry of a bin, shift lookup, skipped samples, mean   deform_psroi_pooling_cuda.cu:59-147 (forward kernel)
  bilinear taps (floor / ceil corners)                      deform_psroi_pooling_cuda.cu:34-57
  gradients of the map and of the shifts                    deform_psroi_pooling_cuda.cu:149-262 (backward kernel)
  shapes, num_classes / channels_each_class                 deform_psroi_pooling_cuda.cu:264-302
Parity pin: the reference holds no vectors for this op and its CUDA cannot run here; its own self-checks
(DeformableConvolutionV2PyTorch/test.py:437-468 zero shifts == no_trans, :471-505 gradcheck) are restated in
tests/test_psroi_oracle.py (this file's backward against finite differences of its forward in float64, a hand-worked
known answer on the self-check's block image) and tests/test_psroi_gpu.py (the HIP kernels against this file)."""
import numpy as np


def _round_half_away(x):
    a = np.abs(x)
    f = np.floor(a)
    return np.copysign(f + ((a - f) >= x.dtype.type(0.5)), x).astype(x.dtype)


def _bins(data, rois, trans, no_trans, spatial_scale, output_dim, group_size, pooled, part, trans_std):
    T = data.dtype.type
    B, C, H, W = data.shape
    R = rois.shape[0]
    classes = 1 if no_trans else trans.shape[1] // 2
    per_class = output_dim if no_trans else output_dim // classes
    n, ctop, ph, pw = np.meshgrid(np.arange(R), np.arange(output_dim), np.arange(pooled), np.arange(pooled), indexing="ij")
    r = rois.astype(data.dtype)
    bi = r[:, 0].astype(np.int64)[n]
    start_w = (_round_half_away(r[:, 1]) * T(spatial_scale) - T(0.5))[n]
    start_h = (_round_half_away(r[:, 2]) * T(spatial_scale) - T(0.5))[n]
    end_w = ((_round_half_away(r[:, 3]) + T(1)) * T(spatial_scale) - T(0.5))[n]
    end_h = ((_round_half_away(r[:, 4]) + T(1)) * T(spatial_scale) - T(0.5))[n]
    roi_w = np.maximum(end_w - start_w, T(0.1))
    roi_h = np.maximum(end_h - start_h, T(0.1))
    bin_h, bin_w = roi_h / T(pooled), roi_w / T(pooled)
    part_h = np.floor(ph.astype(data.dtype) / T(pooled) * T(part)).astype(np.int64)
    part_w = np.floor(pw.astype(data.dtype) / T(pooled) * T(part)).astype(np.int64)
    cls = ctop // per_class
    if no_trans:
        tx = ty = np.zeros(n.shape, data.dtype)
    else:
        tx = trans[n, 2 * cls, part_h, part_w] * T(trans_std)
        ty = trans[n, 2 * cls + 1, part_h, part_w] * T(trans_std)
    wstart = pw.astype(data.dtype) * bin_w + start_w
    wstart = wstart + tx * roi_w
    hstart = ph.astype(data.dtype) * bin_h + start_h
    hstart = hstart + ty * roi_h
    gw = np.clip(np.floor(pw.astype(data.dtype) * T(group_size) / T(pooled)).astype(np.int64), 0, group_size - 1)
    gh = np.clip(np.floor(ph.astype(data.dtype) * T(group_size) / T(pooled)).astype(np.int64), 0, group_size - 1)
    c = (ctop * group_size + gh) * group_size + gw
    live = (bi >= 0) & (bi < B)
    return dict(n=n, cls=cls, part_h=part_h, part_w=part_w, bi=np.where(live, bi, 0), live=live, c=c, wstart=wstart, hstart=hstart,
                bin_w=bin_w, bin_h=bin_h, roi_w=roi_w, roi_h=roi_h)


def _samples(data, b, sample_per_part):
    """Yields, per (ih, iw) in the reference's loop order: inside mask, the four tap coordinates and the two fractions."""
    T = data.dtype.type
    H, W = data.shape[2:]
    sub_h, sub_w = b["bin_h"] / T(sample_per_part), b["bin_w"] / T(sample_per_part)
    for ih in range(sample_per_part):
        for iw in range(sample_per_part):
            w = b["wstart"] + T(iw) * sub_w
            h = b["hstart"] + T(ih) * sub_h
            inside = ~((w < T(-0.5)) | (w > T(W - 0.5)) | (h < T(-0.5)) | (h > T(H - 0.5))) & b["live"]
            w = np.minimum(np.maximum(w, T(0)), T(W - 1))
            h = np.minimum(np.maximum(h, T(0)), T(H - 1))
            x0, x1 = np.floor(w).astype(np.int64), np.ceil(w).astype(np.int64)
            y0, y1 = np.floor(h).astype(np.int64), np.ceil(h).astype(np.int64)
            yield inside, x0, x1, y0, y1, w - x0.astype(data.dtype), h - y0.astype(data.dtype)


def deform_psroi_pooling_forward(data, rois, trans, no_trans, spatial_scale, output_dim, group_size, pooled_size, part_size,
                                 sample_per_part, trans_std):
    """-> (out, top_count), both [R, output_dim, pooled, pooled] in data's dtype."""
    data = np.ascontiguousarray(data)
    T = data.dtype.type
    assert data.shape[1] == output_dim * group_size * group_size
    b = _bins(data, rois, trans, no_trans, spatial_scale, output_dim, group_size, pooled_size, part_size, trans_std)
    total = np.zeros(b["n"].shape, data.dtype)
    count = np.zeros(b["n"].shape, np.int64)
    for inside, x0, x1, y0, y1, dx, dy in _samples(data, b, sample_per_part):
        v11, v12 = data[b["bi"], b["c"], y0, x0], data[b["bi"], b["c"], y1, x0]
        v21, v22 = data[b["bi"], b["c"], y0, x1], data[b["bi"], b["c"], y1, x1]
        val = (T(1) - dx) * (T(1) - dy) * v11 + (T(1) - dx) * dy * v12 + dx * (T(1) - dy) * v21 + dx * dy * v22
        total = np.where(inside, total + val, total)
        count += inside
    out = np.where(count == 0, T(0), total / np.maximum(count, 1).astype(data.dtype))
    return out.astype(data.dtype), count.astype(data.dtype)


def deform_psroi_pooling_backward(grad_out, data, rois, trans, top_count, no_trans, spatial_scale, output_dim, group_size, pooled_size,
                                  part_size, sample_per_part, trans_std):
    """-> (grad_data, grad_trans); sums accumulated in float64 (the kernel's atomics have no order to restate)."""
    data = np.ascontiguousarray(data)
    T = data.dtype.type
    b = _bins(data, rois, trans, no_trans, spatial_scale, output_dim, group_size, pooled_size, part_size, trans_std)
    gdata = np.zeros(data.shape, np.float64)
    gtrans = np.zeros((0,) if no_trans else trans.shape, np.float64)
    has = top_count > 0
    diff = np.where(has, grad_out / np.where(has, top_count, 1), 0).astype(data.dtype)
    for inside, x0, x1, y0, y1, dx, dy in _samples(data, b, sample_per_part):
        m = inside & has
        bi, c = b["bi"][m], b["c"][m]
        X0, X1, Y0, Y1, DX, DY, D = x0[m], x1[m], y0[m], y1[m], dx[m], dy[m], diff[m]
        np.add.at(gdata, (bi, c, Y0, X0), (T(1) - DX) * (T(1) - DY) * D)
        np.add.at(gdata, (bi, c, Y1, X0), (T(1) - DX) * DY * D)
        np.add.at(gdata, (bi, c, Y0, X1), DX * (T(1) - DY) * D)
        np.add.at(gdata, (bi, c, Y1, X1), DX * DY * D)
        if no_trans:
            continue
        u00, u01, u10, u11 = data[bi, c, Y0, X0], data[bi, c, Y1, X0], data[bi, c, Y0, X1], data[bi, c, Y1, X1]
        sx = (u11 * DY + u10 * (T(1) - DY) - u01 * DY - u00 * (T(1) - DY)) * T(trans_std) * D * b["roi_w"][m]
        sy = (u11 * DX + u01 * (T(1) - DX) - u10 * DX - u00 * (T(1) - DX)) * T(trans_std) * D * b["roi_h"][m]
        np.add.at(gtrans, (b["n"][m], 2 * b["cls"][m], b["part_h"][m], b["part_w"][m]), sx)
        np.add.at(gtrans, (b["n"][m], 2 * b["cls"][m] + 1, b["part_h"][m], b["part_w"][m]), sy)
    return gdata.astype(data.dtype), gtrans.astype(data.dtype)
