"""CPU: the iou3d oracle (reference box_overlap restated) against closed forms and an independent float64
polygon clip — the anchors that stand in for the reference's missing test vectors (oracle header)."""
import ctypes

import numpy as np

import oracle
from boxes_util import exact_overlap, random_boxes


def test_deterministic_trig_accuracy():
    """include/fv2p_math.h: |err| <= 1.5e-7 for sin/cos on |x| <= 8192, atan2 within 4 ulp."""
    L = oracle.lib()
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.uniform(-8, 8, 100000), rng.uniform(-8192, 8192, 100000), [0, np.pi, -np.pi, np.pi / 2]]).astype(np.float32)
    y = rng.uniform(-8, 8, x.size).astype(np.float32)
    s, c, a = np.empty_like(x), np.empty_like(x), np.empty_like(x)
    P = ctypes.POINTER(ctypes.c_float)
    L.oracle_math_eval(x.ctypes.data_as(P), y.ctypes.data_as(P), ctypes.c_int64(x.size), s.ctypes.data_as(P), c.ctypes.data_as(P), a.ctypes.data_as(P))
    xd = x.astype(np.float64)
    assert np.abs(s - np.sin(xd)).max() < 1.5e-7 and np.abs(c - np.cos(xd)).max() < 1.5e-7
    ref = np.arctan2(y.astype(np.float64), xd)
    assert (np.abs(a - ref) / np.spacing(np.abs(ref).astype(np.float32))).max() < 4


def test_closed_form_overlaps():
    a = np.array([[0, 0, 0, 4, 2, 1.5, 0]], np.float32)
    b = np.array([[1, 0.5, 0, 4, 2, 1.5, 0],        # axis-aligned shift: overlap 3 x 1.5
                  [0, 0, 0, 2, 2, 1.5, np.pi / 4],   # diamond inside a 4x2 box: clipped hexagon
                  [10, 10, 0, 4, 2, 1.5, 0.3],       # disjoint
                  [0, 0, 0, 4, 2, 1.5, np.pi / 2]],  # same box turned 90 degrees: 2 x 2 square
                 np.float32)
    ov = oracle.boxes_bev(a, b, "overlap")[0]
    assert abs(ov[0] - 4.5) < 1e-4 and ov[2] == 0 and abs(ov[3] - 4.0) < 1e-3
    assert abs(ov[1] - exact_overlap(a[0], b[1])) < 1e-3
    iou = oracle.boxes_bev(a, a, "iou")
    assert abs(iou[0, 0] - 1.0) < 1e-5


def test_random_pairs_match_float64_clip():
    boxes = random_boxes(1, 120)
    ov = oracle.boxes_bev(boxes[:60], boxes[60:], "overlap")
    iou = oracle.boxes_bev(boxes[:60], boxes[60:], "iou")
    worst = 0.0
    for i in range(60):
        for j in range(60):
            worst = max(worst, abs(float(ov[i, j]) - exact_overlap(boxes[i], boxes[60 + j])))
    # the reference's corner test uses a 1e-2 margin (iou3d_nms_kernel.cu:53): allow a few 1e-2 * edge length
    assert worst < 0.08
    assert (iou >= 0).all() and (iou <= 1.0 + 1e-5).all()
    assert (ov > 0).mean() > 0.02  # the sample does contain overlapping pairs


def test_nms_properties():
    boxes = random_boxes(2, 600)
    scores = np.random.default_rng(2).permutation(600).astype(np.float32)
    keep = oracle.nms(boxes, scores, 0.3)
    assert len(set(keep.tolist())) == len(keep) and len(keep) < 600
    assert (np.diff(scores[keep]) < 0).all()  # survivors come out in descending score order
    kb = boxes[keep]
    iou = oracle.boxes_bev(kb, kb, "iou")
    assert (np.triu(iou, 1) <= 0.3).all()  # no survivor is suppressed by an earlier survivor
    # idempotence: NMS of the survivors keeps all of them
    assert len(oracle.nms(kb, scores[keep], 0.3)) == len(keep)


def test_all_core_forms_equal_the_serial_ones():
    """oracle_boxes_bev_mt / oracle_nms_mt (bench.py's all-core B2 / B3 baselines): OpenMP over the rows of the
    pair matrix changes who computes a pair, not what is computed - values and survivors equal the serial forms bit for bit."""
    import oracle
    from fv2p_harness import synth
    b = synth.proposal_boxes(3, 300)
    assert np.array_equal(oracle.boxes_bev(b, b, "iou"), oracle.boxes_bev(b, b, "iou", threads=4))
    assert np.array_equal(oracle.boxes_bev(b[:50], b, "overlap"), oracle.boxes_bev(b[:50], b, "overlap", threads=3))
    for tight, thr in ((True, 0.8), (False, 0.8), (True, 0.1)):
        bx = synth.proposal_boxes(1, 1500, tight=tight)
        s = np.random.default_rng(0).standard_normal(1500).astype(np.float32)
        assert np.array_equal(oracle.nms(bx, s, thr), oracle.nms(bx, s, thr, threads=4))
    assert len(oracle.nms(b[:0], np.zeros(0, np.float32), 0.5, threads=2)) == 0
