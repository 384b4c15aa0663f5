"""CPU checks of the deformable PSROI pooling restatement (oracle/psroi_oracle.py): the reference's own
self-checks (DeformableConvolutionV2PyTorch/test.py:437-468, :471-505) restated without a GPU, a closed-form known answer
on the self-check's block image, and the backward pass against finite differences of the forward pass in float64."""
import numpy as np

from oracle import psroi_oracle as ps


def _case(seed, dtype=np.float32, batch=2, channels=3, size=5, rois=4, pooled=3, classes=1):
    """Shapes and RoI statistics of the reference's check_gradient_dpooling (test.py:471-483)."""
    rng = np.random.default_rng(seed)
    data = (rng.standard_normal((batch, channels, size, size)) * 0.5).astype(dtype)
    bi = rng.integers(0, batch, (rois, 1)).astype(dtype)
    x, y = rng.random((rois, 1)) * 15, rng.random((rois, 1)) * 15
    w, h = rng.random((rois, 1)) * 10, rng.random((rois, 1)) * 10
    boxes = np.concatenate([bi, x, y, x + w, y + h], 1).astype(dtype)
    trans = rng.standard_normal((rois, 2 * classes, pooled, pooled)).astype(dtype)
    return data, boxes, trans


def test_block_image_known_answer():
    """check_pooling_zero_offset's image (test.py:439-445): a 10 x 10 block of ones, RoI (65, 65, 103, 103) at scale 1/4.  Every
    sample lies on the map, the image is a product of two box functions, so a bin is the product of two means of the
    piecewise-linear interpolant t -> clip(min(t - 15, 26 - t), 0, 1): computed here in float64 from that formula."""
    data = np.zeros((2, 16, 64, 64), np.float32)
    data[0, :, 16:26, 16:26] = 1.0
    data[1, :, 10:20, 20:30] = 2.0
    rois = np.array([[0, 65, 65, 103, 103], [1, 81, 41, 119, 79]], np.float32)
    out, cnt = ps.deform_psroi_pooling_forward(data, rois, None, True, 0.25, 16, 1, 7, 7, 4, 0.0)
    assert out.shape == (2, 16, 7, 7) and np.all(cnt == 16)

    def line(lo, x1, x2):   # mean over the 4 samples of each of the 7 bins of the interpolated box function [lo, lo + 10)
        start, end = x1 * 0.25 - 0.5, (x2 + 1) * 0.25 - 0.5
        t = start + (np.arange(7)[:, None] * 4 + np.arange(4)[None, :]) * ((end - start) / 28)
        return np.clip(np.minimum(t - (lo - 1), (lo + 10) - t), 0, 1).mean(1)
    want0 = np.outer(line(16, 65, 103), line(16, 65, 103))
    want1 = 2.0 * np.outer(line(10, 41, 79), line(20, 81, 119))
    assert np.abs(out[0] - want0).max() < 2e-6 and np.abs(out[1] - want1).max() < 4e-6     # fp32 sample positions vs float64
    assert 0.9 < out[0].mean() < 1.0       # the printed means of the reference's check: just under 1 and 2
    # second half of the self-check: zero shifts pool like no_trans
    dout, _ = ps.deform_psroi_pooling_forward(data, rois, np.zeros((20, 2, 7, 7), np.float32), False, 0.25, 16, 1, 7, 7, 4, 0.0)
    assert np.array_equal(dout, out)


def test_shift_semantics():
    data, rois, trans = _case(0, size=24, channels=4, classes=2)
    plain, _ = ps.deform_psroi_pooling_forward(data, rois, None, True, 0.25, 4, 1, 3, 3, 4, 0.1)
    frozen, _ = ps.deform_psroi_pooling_forward(data, rois, trans, False, 0.25, 4, 1, 3, 3, 4, 0.0)     # trans_std = 0: shifts ignored
    assert np.array_equal(plain, frozen)
    moved, _ = ps.deform_psroi_pooling_forward(data, rois, trans, False, 0.25, 4, 1, 3, 3, 4, 0.1)
    assert np.abs(moved - plain).max() > 1e-3
    # the two shift classes address channel halves: changing class 1's planes leaves channels 0-1 alone
    t2 = trans.copy()
    t2[:, 2:] += 1.0
    moved2, _ = ps.deform_psroi_pooling_forward(data, rois, t2, False, 0.25, 4, 1, 3, 3, 4, 0.1)
    assert np.array_equal(moved2[:, :2], moved[:, :2]) and np.abs(moved2[:, 2:] - moved[:, 2:]).max() > 1e-3


def test_rois_off_the_map_and_outside_the_batch():
    data, _, _ = _case(1, size=8)
    rois = np.array([[0, -40, -40, -30, -30],      # wholly off the map: nothing sampled
                     [1, 20, 20, 60, 60],          # hangs over the lower right edge: partial counts
                     [5, 0, 0, 10, 10],            # batch index outside the batch: zeros (the reference would read beyond the map)
                     [0, 2.5, 3.5, 2.5, 3.5]], np.float32)   # corners round half away from zero: (3, 4, 3, 4)
    out, cnt = ps.deform_psroi_pooling_forward(data, rois, None, True, 0.25, 3, 1, 3, 3, 4, 0.0)
    assert np.all(cnt[0] == 0) and np.all(out[0] == 0)
    assert cnt[1].max() == 16 and cnt[1].min() < 16
    assert np.all(cnt[2] == 0) and np.all(out[2] == 0)
    same, _ = ps.deform_psroi_pooling_forward(data, np.array([[0, 3, 4, 3, 4]], np.float32), None, True, 0.25, 3, 1, 3, 3, 4, 0.0)
    assert np.array_equal(out[3], same[0])


def test_group_size_reads_position_sensitive_channels():
    """group_size g: bin (ph, pw) of output channel c reads input channel (c * g + gh) * g + gw (deform_psroi_pooling_cuda.cu:118-136)."""
    rng = np.random.default_rng(2)
    g, out_dim, pooled = 2, 3, 4
    data = rng.standard_normal((1, out_dim * g * g, 12, 12)).astype(np.float32)
    rois = np.array([[0, 4, 4, 40, 40]], np.float32)
    out, _ = ps.deform_psroi_pooling_forward(data, rois, None, True, 0.25, out_dim, g, pooled, pooled, 2, 0.0)
    for c in range(out_dim):
        for gh in range(g):
            for gw in range(g):
                plane = data[:, (c * g + gh) * g + gw][:, None]
                one, _ = ps.deform_psroi_pooling_forward(plane, rois, None, True, 0.25, 1, 1, pooled, pooled, 2, 0.0)
                assert np.array_equal(out[0, c, gh * 2:(gh + 1) * 2, gw * 2:(gw + 1) * 2], one[0, 0, gh * 2:(gh + 1) * 2, gw * 2:(gw + 1) * 2])


def test_backward_matches_finite_differences():
    """check_gradient_dpooling (test.py:471-505) on the restatement: float64, central differences of sum(out * g)."""
    data, rois, trans = _case(3, np.float64)
    conf = (False, 0.25, 3, 1, 3, 3, 4, 0.1)
    rng = np.random.default_rng(4)
    out, cnt = ps.deform_psroi_pooling_forward(data, rois, trans, *conf)
    g = rng.standard_normal(out.shape)
    gdata, gtrans = ps.deform_psroi_pooling_backward(g, data, rois, trans, cnt, *conf)
    assert np.abs(gdata).max() > 0 and np.abs(gtrans).max() > 0

    def f(d, t):
        return float((ps.deform_psroi_pooling_forward(d, rois, t, *conf)[0] * g).sum())
    eps = 1e-6
    for arr, grad, which in ((data, gdata, 0), (trans, gtrans, 1)):
        num = np.zeros_like(arr)
        it = np.nditer(arr, flags=["multi_index"])
        for _ in it:
            i = it.multi_index
            hi, lo = arr.copy(), arr.copy()
            hi[i] += eps
            lo[i] -= eps
            num[i] = (f(hi, trans) - f(lo, trans)) / (2 * eps) if which == 0 else (f(data, hi) - f(data, lo)) / (2 * eps)
        assert np.abs(num - grad).max() < 1e-5 * max(1.0, np.abs(grad).max()), which
    # without shifts only the map has a gradient, and it is the same accumulation
    out0, cnt0 = ps.deform_psroi_pooling_forward(data, rois, None, True, *conf[1:])
    g0, t0 = ps.deform_psroi_pooling_backward(g, data, rois, None, cnt0, True, *conf[1:])
    assert t0.size == 0 and np.isfinite(g0).all()
