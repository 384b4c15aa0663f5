"""`pcdet.ops.iou3d_nms.iou3d_nms_cuda` — the five functions the reference binds
(pcdet/ops/iou3d_nms/src/iou3d_nms_api.cpp:11-17) with the same argument conventions: caller-allocated
outputs are filled in place, `keep` of the NMS calls is a CPU LongTensor, the return value is 1 or the count.
Errors raise (the reference prints and exit(-1)s)."""
import torch

import fv2p_native as _nat


def _check(t, name, cuda=True):
    if not isinstance(t, torch.Tensor) or not t.is_contiguous() or (not cuda and t.is_cuda):
        raise _nat.Fv2pError(f"{name} must be a contiguous {'CUDA' if cuda else 'CPU'} tensor")
    if cuda:
        _nat.require_cuda(t)


def boxes_overlap_bev_gpu(boxes_a, boxes_b, ans_overlap):
    for t, n in ((boxes_a, "boxes_a"), (boxes_b, "boxes_b"), (ans_overlap, "ans_overlap")):
        _check(t, n)
    with _nat.device_guard(boxes_a.device):
        _nat.call("fv2p_boxes_overlap_bev", boxes_a, boxes_a.shape[0], boxes_b, boxes_b.shape[0], ans_overlap, _nat.stream())
    return 1


def boxes_iou_bev_gpu(boxes_a, boxes_b, ans_iou):
    for t, n in ((boxes_a, "boxes_a"), (boxes_b, "boxes_b"), (ans_iou, "ans_iou")):
        _check(t, n)
    with _nat.device_guard(boxes_a.device):
        _nat.call("fv2p_boxes_iou_bev", boxes_a, boxes_a.shape[0], boxes_b, boxes_b.shape[0], ans_iou, _nat.stream())
    return 1


def _nms(boxes, keep, thresh, normal):
    _check(boxes, "boxes")
    if keep.is_cuda or keep.dtype != torch.int64:
        raise _nat.Fv2pError("keep must be a CPU LongTensor (iou3d_nms.cpp:96-97)")
    n = boxes.shape[0]
    dev_keep, cnt = nms_device(boxes, thresh, normal)
    num = int(cnt.item())
    keep[:num] = dev_keep[:num].cpu()
    return num


def nms_device(boxes, thresh, normal=False):
    """Device-resident variant: returns (keep int64 [N] on the GPU, count int32 [1] on the GPU) without any
    host synchronisation — used by iou3d_nms_utils.nms_gpu to keep survivors on the device."""
    n = boxes.shape[0]
    keep = torch.empty((max(n, 1),), dtype=torch.int64, device=boxes.device)
    cnt = torch.zeros((1,), dtype=torch.int32, device=boxes.device)
    with _nat.device_guard(boxes.device):
        nb = _nat.lib().fv2p_nms_ws_bytes(n)
        ws = _nat.workspace(nb, boxes.device)
        _nat.call("fv2p_nms", boxes, n, float(thresh), int(bool(normal)), keep, cnt, ws, ws.numel(), _nat.stream())
    return keep, cnt


def nms_batch_device(boxes, thresh, max_keep=0, normal=False):
    """boxes (B, n, 7), each sample sorted by descending score -> (keep (B, K) int64, count (B) int32) on the device, with
    K = min(max_keep, n) survivors per sample at most (max_keep <= 0: all of them, K = n).  One launch sequence for the whole
    batch, no host synchronisation; rows of `keep` past `count` are undefined."""
    _check(boxes, "boxes")
    b, n, _ = boxes.shape
    k = min(max_keep, n) if max_keep > 0 else n
    keep = torch.empty((b, max(k, 1)), dtype=torch.int64, device=boxes.device)
    cnt = torch.zeros((b,), dtype=torch.int32, device=boxes.device)
    with _nat.device_guard(boxes.device):
        ws = _nat.workspace(_nat.lib().fv2p_nms_batch_ws_bytes(b, n, int(max_keep)), boxes.device)
        _nat.call("fv2p_nms_batch", boxes, b, n, float(thresh), int(bool(normal)), int(max_keep), keep, keep.shape[1], cnt, ws, ws.numel(),
                  _nat.stream())
    return keep, cnt


def nms_gpu(boxes, keep, nms_overlap_thresh):
    return _nms(boxes, keep, nms_overlap_thresh, False)


def nms_normal_gpu(boxes, keep, nms_overlap_thresh):
    return _nms(boxes, keep, nms_overlap_thresh, True)


def boxes_iou_bev_cpu(boxes_a, boxes_b, ans_iou):
    for t, n in ((boxes_a, "boxes_a"), (boxes_b, "boxes_b"), (ans_iou, "ans_iou")):
        _check(t, n, cuda=False)
    _nat.call("fv2p_boxes_iou_bev_cpu", boxes_a, boxes_a.shape[0], boxes_b, boxes_b.shape[0], ans_iou)
    return 1
