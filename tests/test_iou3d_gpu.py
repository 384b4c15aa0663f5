"""GPU parity of iou3d_nms (A16) through pcdet.ops.iou3d_nms against the oracle restatement.

Bar: NMS survivor indices bit-exact; overlaps / IoUs bit-exact too (same fp32 operation sequence and the same
deterministic sin/cos/atan2 on both sides) — asserted with array_equal, tolerance 0."""
import numpy as np
import pytest
import torch

import oracle
from boxes_util import random_boxes
from pcdet.ops.iou3d_nms import iou3d_nms_cuda, iou3d_nms_utils

pytestmark = pytest.mark.gpu


def test_pairwise_overlap_and_iou(gpu):
    a, b = random_boxes(3, 512), random_boxes(4, 300)
    ta, tb = torch.from_numpy(a).to(gpu), torch.from_numpy(b).to(gpu)
    iou = iou3d_nms_utils.boxes_iou_bev(ta, tb)
    assert np.array_equal(iou.cpu().numpy(), oracle.boxes_bev(a, b, "iou"))
    ov = torch.zeros((512, 300), device=gpu)
    iou3d_nms_cuda.boxes_overlap_bev_gpu(ta, tb, ov)
    assert np.array_equal(ov.cpu().numpy(), oracle.boxes_bev(a, b, "overlap"))
    i3 = iou3d_nms_utils.boxes_iou3d_gpu(ta, tb)
    assert np.abs(i3.cpu().numpy() - oracle.boxes_iou3d(a, b)).max() < 1e-6
    bb = iou3d_nms_utils.batch_boxes_iou3d_gpu(ta[None, :64], tb[None, :32])
    assert bb.shape == (1, 64, 1)
    # CPU entry point of the extension (boxes_iou_bev_cpu): host tensors / numpy in, same numbers
    assert np.array_equal(iou3d_nms_utils.boxes_bev_iou_cpu(a[:50], b[:40]), oracle.boxes_bev(a[:50], b[:40], "iou"))


@pytest.mark.parametrize("n,thresh,pre,normal", [(9000, 0.8, None, False), (4096, 0.1, None, False), (9000, 0.7, 4096, False),
                                                 (1000, 0.5, None, True), (65, 0.3, None, False), (64, 0.3, None, False), (1, 0.5, None, False)])
def test_nms_survivors_bit_exact(gpu, n, thresh, pre, normal):
    boxes = random_boxes(n, n)
    scores = np.random.default_rng(n + 1).permutation(n).astype(np.float32)  # distinct scores: sort order is unambiguous
    fn = iou3d_nms_utils.nms_normal_gpu if normal else iou3d_nms_utils.nms_gpu
    kw = {} if normal else {"pre_maxsize": pre}
    keep, none = fn(torch.from_numpy(boxes).to(gpu), torch.from_numpy(scores).to(gpu), thresh, **kw)
    assert none is None and keep.dtype == torch.int64 and keep.is_cuda
    ref = oracle.nms(boxes, scores, thresh, pre_maxsize=pre, normal=normal)
    assert np.array_equal(keep.cpu().numpy(), ref)


def test_nms_ext_convention_and_errors(gpu):
    boxes = random_boxes(7, 500)
    order = np.argsort(-np.arange(500, dtype=np.float32), kind="stable")
    tb = torch.from_numpy(boxes).to(gpu)
    keep = torch.LongTensor(500)
    num = iou3d_nms_cuda.nms_gpu(tb, keep, 0.5)  # reference convention: CPU LongTensor filled, count returned
    ref = oracle.nms(boxes, -np.arange(500, dtype=np.float32), 0.5)
    assert num == len(ref) and np.array_equal(keep[:num].numpy(), ref)
    with pytest.raises(Exception):
        iou3d_nms_cuda.nms_gpu(tb.cpu(), keep, 0.5)       # boxes must be on the GPU
    with pytest.raises(Exception):
        iou3d_nms_cuda.nms_gpu(tb, keep.to(gpu), 0.5)      # keep must be a CPU tensor
    empty = iou3d_nms_utils.nms_gpu(torch.zeros((0, 7), device=gpu), torch.zeros((0,), device=gpu), 0.5)[0]
    assert empty.numel() == 0


@pytest.mark.parametrize("n,thresh,max_keep,normal", [(9000, 0.8, 512, False), (9000, 0.1, 512, False), (4096, 0.05, 500, False),
                                                      (3000, 0.3, 0, False), (700, 0.5, 100, True), (65, 0.3, 64, False), (40, 0.3, 512, False)])
def test_batched_truncated_nms_is_the_head_of_the_full_list(gpu, n, thresh, max_keep, normal):
    """fv2p_nms_batch: three samples in one launch sequence; with max_keep the greedy pass stops early, and what it returns
    is exactly the first max_keep survivors of the full pass (low thresholds force several mask chunks)."""
    sets = [random_boxes(100 + s, n) for s in range(3)]
    keep, cnt = iou3d_nms_cuda.nms_batch_device(torch.from_numpy(np.stack(sets)).to(gpu), thresh, max_keep, normal)
    keep, cnt = keep.cpu().numpy(), cnt.cpu().numpy()
    for s in range(3):
        ref = oracle.nms(sets[s], -np.arange(n, dtype=np.float32), thresh, normal=normal)
        if max_keep > 0:
            ref = ref[:max_keep]
        assert cnt[s] == len(ref)
        assert np.array_equal(keep[s, :cnt[s]], ref)


def test_nms_gpu_honours_the_callers_post_maxsize(gpu):
    """The reference's class_agnostic_nms passes its whole config as keywords (model_nms_utils.py:14-16) and then keeps
    selected[:NMS_POST_MAXSIZE]: with the keyword present nms_gpu returns exactly that head."""
    boxes = random_boxes(11, 5000)
    scores = np.random.default_rng(5).permutation(5000).astype(np.float32)
    tb, ts = torch.from_numpy(boxes).to(gpu), torch.from_numpy(scores).to(gpu)
    cfg = dict(NMS_TYPE="nms_gpu", MULTI_CLASSES_NMS=False, NMS_PRE_MAXSIZE=4096, NMS_POST_MAXSIZE=100, NMS_THRESH=0.4)
    keep, _ = iou3d_nms_utils.nms_gpu(tb, ts, 0.4, pre_maxsize=4096, **{k: v for k, v in cfg.items() if k != "NMS_PRE_MAXSIZE"})
    ref = oracle.nms(boxes, scores, 0.4, pre_maxsize=4096)[:100]
    assert np.array_equal(keep.cpu().numpy(), ref)


def iou3d_composition(boxes_a, boxes_b):
    """iou3d_nms_utils.py:454-491 as the reference composes it: boxes_overlap_bev_gpu + a dozen torch ops."""
    a_max = (boxes_a[:, 2] + boxes_a[:, 5] / 2).view(-1, 1)
    a_min = (boxes_a[:, 2] - boxes_a[:, 5] / 2).view(-1, 1)
    b_max = (boxes_b[:, 2] + boxes_b[:, 5] / 2).view(1, -1)
    b_min = (boxes_b[:, 2] - boxes_b[:, 5] / 2).view(1, -1)
    overlaps_bev = torch.zeros((boxes_a.shape[0], boxes_b.shape[0]), dtype=torch.float32, device=boxes_a.device)
    iou3d_nms_cuda.boxes_overlap_bev_gpu(boxes_a.contiguous(), boxes_b.contiguous(), overlaps_bev)
    overlaps_h = torch.clamp(torch.min(a_max, b_max) - torch.max(a_min, b_min), min=0)
    overlaps_3d = overlaps_bev * overlaps_h
    vol_a = (boxes_a[:, 3] * boxes_a[:, 4] * boxes_a[:, 5]).view(-1, 1)
    vol_b = (boxes_b[:, 3] * boxes_b[:, 4] * boxes_b[:, 5]).view(1, -1)
    iou3d = overlaps_3d / torch.clamp(vol_a + vol_b - overlaps_3d, min=1e-6)
    return iou3d.clamp(0, 1)


@pytest.mark.gpu
def test_batched_iou3d_equals_the_per_sample_composition(gpu):
    """fv2p_boxes_iou3d_batch == the reference's composition per sample (bit for bit: the same float operations in the same order),
    with ground-truth rows of 8 values (class id last) and zero-padded rows; boxes_iou3d_gpu itself is that kernel with a batch of one."""
    import fv2p_native
    a = torch.from_numpy(np.stack([random_boxes(3 + i, 200, spread=8.0) for i in range(3)])).to(gpu)
    g7 = np.stack([random_boxes(30 + i, 24, spread=8.0) for i in range(3)])
    g8 = np.concatenate([g7, np.ones((3, 24, 1), np.float32)], 2)
    g8[:, 20:] = 0
    b = torch.from_numpy(g8).to(gpu)
    out = torch.empty((3, 200, 24), device=gpu)
    fv2p_native.call("fv2p_boxes_iou3d_batch", a, 3, 200, b, 24, 8, out, fv2p_native.stream())
    for i in range(3):
        want = iou3d_composition(a[i], b[i, :, :7].contiguous())
        assert torch.equal(out[i], want)
        assert torch.equal(iou3d_nms_utils.boxes_iou3d_gpu(a[i], b[i, :, :7].contiguous()), want)
    assert float(out.max()) > 0.05
    assert iou3d_nms_utils.boxes_iou3d_gpu(a[0][:0], b[0, :, :7].contiguous()).shape == (0, 24)
