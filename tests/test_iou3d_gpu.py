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
