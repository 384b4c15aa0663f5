"""3-D IoU / rotated NMS helpers — the call surface of the reference's
pcdet/ops/iou3d_nms/iou3d_nms_utils.py:418-562 (boxes_bev_iou_cpu, boxes_iou_bev, boxes_iou3d_gpu, nms_gpu,
nms_normal_gpu, batch_boxes_iou3d_gpu) on the HIP kernels of libfv2p_ops.

The fork's experimental pure-python NMS variants (soft_nms_torch, iou_weighted_nms_cpu, matched_boxes_iou3d_cpu,
:16-415) need shapely and are referenced by no config: they are out of scope here."""
import torch

from ...utils import common_utils
from . import iou3d_nms_cuda


def boxes_bev_iou_cpu(boxes_a, boxes_b):
    """(N,7) x (M,7) CPU tensors / numpy -> (N,M) rotated BEV IoU, computed on the host."""
    boxes_a, is_numpy = common_utils.check_numpy_to_torch(boxes_a)
    boxes_b, is_numpy = common_utils.check_numpy_to_torch(boxes_b)
    assert not (boxes_a.is_cuda or boxes_b.is_cuda), 'Only support CPU tensors'
    assert boxes_a.shape[1] == 7 and boxes_b.shape[1] == 7
    ans_iou = boxes_a.new_zeros(torch.Size((boxes_a.shape[0], boxes_b.shape[0])))
    iou3d_nms_cuda.boxes_iou_bev_cpu(boxes_a.contiguous(), boxes_b.contiguous(), ans_iou)
    return ans_iou.numpy() if is_numpy else ans_iou


def boxes_iou_bev(boxes_a, boxes_b):
    assert boxes_a.shape[1] == boxes_b.shape[1] == 7
    ans_iou = torch.zeros((boxes_a.shape[0], boxes_b.shape[0]), dtype=torch.float32, device=boxes_a.device)
    iou3d_nms_cuda.boxes_iou_bev_gpu(boxes_a.contiguous(), boxes_b.contiguous(), ans_iou)
    return ans_iou


def boxes_iou3d_gpu(boxes_a, boxes_b):
    """(N,7) x (M,7) -> (N,M) 3-D IoU = BEV overlap x height overlap / union, clamped to [0,1] (:454-491)."""
    assert boxes_a.shape[1] == boxes_b.shape[1] == 7
    if boxes_a.is_cuda and boxes_a.dtype == torch.float32 and boxes_b.dtype == torch.float32 and boxes_a.shape[0] > 0 and boxes_b.shape[0] > 0:
        # one launch: the same float operations in the same order as the composition below (fv2p_boxes_iou3d_batch with a batch of one;
        # tests/test_iou3d_gpu.py holds the two bit for bit).  The composition is ~20 small launches: 133 us of launch latency for a
        # 512 x 40 problem (profiles/r03_op_roofline.txt).
        import fv2p_native as _nat
        a, b = boxes_a.contiguous(), boxes_b.contiguous()
        iou3d = torch.empty((a.shape[0], b.shape[0]), dtype=torch.float32, device=a.device)
        with _nat.device_guard(a.device):
            _nat.call("fv2p_boxes_iou3d_batch", a, 1, a.shape[0], b, b.shape[0], 7, iou3d, _nat.stream())
        return iou3d
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
    iou3d[iou3d < 0] = 0
    iou3d[iou3d > 1] = 1
    return iou3d


def _nms(boxes, scores, thresh, pre_maxsize, normal, post_maxsize=None):
    assert boxes.shape[1] == 7
    order = scores.sort(0, descending=True)[1]
    if pre_maxsize is not None:
        order = order[:pre_maxsize]
    boxes = boxes[order].contiguous()
    if post_maxsize is not None and post_maxsize > 0:
        # the caller keeps selected[:NMS_POST_MAXSIZE] (model_nms_utils.py:16-20): the greedy pass stops at that survivor
        keep, cnt = iou3d_nms_cuda.nms_batch_device(boxes.unsqueeze(0), thresh, int(post_maxsize), normal)
        keep = keep[0]
    else:
        keep, cnt = iou3d_nms_cuda.nms_device(boxes, thresh, normal)
    num_out = int(cnt.item())  # the survivor count is a tensor shape: one 4-byte D2H instead of the N*N/8-byte mask
    return order[keep[:num_out]].contiguous(), None


def nms_gpu(boxes, scores, thresh, pre_maxsize=None, **kwargs):
    """Rotated NMS: returns (indices into `boxes` of the survivors, in descending score order; None) (:494-509).
    The reference's callers pass their whole NMS config as keyword arguments (model_nms_utils.py:14-16); when it carries
    NMS_POST_MAXSIZE only that many survivors are produced — the caller slices the list to exactly that length."""
    return _nms(boxes, scores, thresh, pre_maxsize, False, kwargs.get("NMS_POST_MAXSIZE"))


def nms_normal_gpu(boxes, scores, thresh, **kwargs):
    """Axis-aligned (heading ignored) NMS (:512-526)."""
    return _nms(boxes, scores, thresh, None, True, kwargs.get("NMS_POST_MAXSIZE"))


def batch_boxes_iou3d_gpu(boxes_a, boxes_b):
    """(B,N,7) x (B,M,7) -> (B,N,1): for every box of a, the max 3-D IoU with the boxes of b (:529-562)."""
    assert boxes_a.shape[0] == boxes_b.shape[0]
    assert boxes_a.shape[-1] == boxes_b.shape[-1]
    out = []
    for i in range(boxes_a.shape[0]):
        iou3d = boxes_iou3d_gpu(boxes_a=boxes_a[i, ...], boxes_b=boxes_b[i, ...])
        max_overlaps, _ = torch.max(iou3d, dim=1)
        out.append(max_overlaps.view(-1, 1))
    return torch.stack(out, dim=0)
