"""Generates tests/golden/pyref_center_targets.npz and pyref_center_losses.npz from the reference's own classes (build container only;
nothing of the reference is copied: the fixtures hold inputs and outputs).

  dense_heads/keypoint_assigner/centertarget_assigner.py:9-209   CenterTargetAssigner (class compiled out of its file)
  utils/center_utils.py:98-155, 250-290                          gaussian_radius, gaussian2D, draw_umich_gaussian, _gather_feat,
                                                                 _transpose_and_gather_feat, _nms, _topk
  dense_heads/center_af_head_template.py:148-598                 build_losses, get_loss and its eight terms, get_max_iou_with_same_class,
                                                                 gthm_based_predicted_boxes_generation, predhm_based_predicted_boxes_generation_ssd
                                                                 (methods compiled out of the class, on a stub head with the yaml's LOSS_CONFIG)
  utils/loss_utils.py                                            CenterNetFocalLoss, CenterNetResLoss, CenterNetRotBinResLoss,
                                                                 SigmoidFocalClassificationLoss, get_corner_loss_mse, get_iouscore_loss_bce,
                                                                 get_rot_binres_loss
  utils/box_utils.py                                             boxes_to_corners_3d, decode_rot_binres

cv2 is not installed here.  The assigner calls it for ONE thing, cv2.fillConvexPoly in draw_seg_mask (the segmentation and height
maps); a stand-in module with a no-op fillConvexPoly lets the class run, and the two maps it would have drawn are left OUT of the
targets fixture (the harness restates them, unpinned).  For the loss fixture the segmentation target is an input like any other.
torch.cuda.FloatTensor (loss_utils.py:395) is torch.FloatTensor while the generator runs; the one CUDA op the losses reach
(boxes_iou3d_gpu) is answered by the oracle.   Run:  python oracle/gen_golden_center_head.py"""
import math
import os
import sys
import types

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_golden_pyref import OUT, by_path, extract, method   # noqa: E402

import oracle  # noqa: E402


class AD(dict):
    __getattr__ = dict.__getitem__


RANGE = np.array([0.0, -8.0, -3.0, 14.0, 8.0, 1.0], np.float32)       # 35 x 40 map at stride 8
VOXEL = [0.05, 0.05, 0.1]
CLASSES = ["Car", "Pedestrian", "Cyclist"]
TARGET_CFG = AD(NAME="CenterTargetAssigner", FEATURE_MAP_STRIDE=8, GAUSSIAN_MINOVERLAP=0.01, GAUSSIAN_MINRADIUS=2, MAX_OBJS=50)
LOSS_CFG = AD(HM_LOSS_CONFIG={"weight": 1.0}, OFFSET_LOSS_CONFIG={"weight": 1.0, "res_func": "l1"}, HEIGHT_LOSS_CONFIG={"weight": 1.0, "res_func": "l1"},
              DIM_LOSS_CONFIG={"weight": 1.0, "res_func": "l1"}, ROT_LOSS_CONFIG={"weight": 1.0, "num_bins": 12}, SEGM_LOSS_CONFIG={"weight": 1.0},
              CORNER_LOSS_CONFIG={"weight": 1.0}, IOUSCORE_LOSS_CONFIG={"weight": 1.0, "iou_fg_thresh": 0.75, "iou_bg_thresh": 0.25})


def gt_batch(rng, b=3, m=14):
    gt = np.zeros((b, m, 8), np.float32)
    sizes = {1: (3.9, 1.6, 1.56), 2: (0.8, 0.6, 1.73), 3: (1.76, 0.6, 1.73)}
    for s in range(b):
        n = [9, 5, 12][s]
        for k in range(n):
            c = int(rng.integers(1, 4))
            dx, dy, dz = np.array(sizes[c]) * rng.uniform(0.8, 1.25, 3)
            gt[s, k] = [rng.uniform(0.5, 13.5), rng.uniform(-7.5, 7.5), rng.uniform(-1.8, -0.6), dx, dy, dz, rng.uniform(-math.pi, math.pi), c]
    gt[0, 3, :2] = [13.99, 7.99]          # centre rounds onto the map's far corner: in range only after the bounds check
    gt[0, 4, :2] = [15.5, 0.0]            # centre outside the map: skipped, its slot stays empty
    gt[1, 2, 3:5] = [0.0, 1.0]            # zero length: skipped
    gt[1, 1] = 0                          # a zero row BETWEEN live rows stays in the list (only trailing zero rows are cut)
    gt[2, 5, :2] = [4.2, 3.4]; gt[2, 6, :2] = [4.6, 3.0]; gt[2, 6, 7] = gt[2, 5, 7]   # overlapping Gaussians of one class
    gt[2, 7, :2] = [0.2, -7.9]            # Gaussian cut by the map border
    return gt


def main():
    rng = np.random.default_rng(77)
    cv2 = types.SimpleNamespace(LINE_AA=16, LINE_4=4, LINE_8=8, fillConvexPoly=lambda img, points, color, lineType: img)
    cu = by_path("ref_common_utils", "utils/common_utils.py")
    bns = {"torch": torch, "np": np, "common_utils": cu}
    extract("utils/box_utils.py", ["boxes_to_corners_3d", "decode_rot_binres"], bns)
    box_utils = types.SimpleNamespace(boxes_to_corners_3d=bns["boxes_to_corners_3d"], decode_rot_binres=bns["decode_rot_binres"])
    cns = {"np": np, "cv2": cv2, "torch": torch, "nn": torch.nn}
    names = ["gaussian_radius", "gaussian2D", "draw_umich_gaussian", "draw_seg_mask", "_gather_feat", "_transpose_and_gather_feat", "_nms", "_topk"]
    extract("utils/center_utils.py", names, cns)
    center_utils = types.SimpleNamespace(**{k: cns[k] for k in names})
    ans = {"np": np, "torch": torch, "math": math, "cv2": cv2, "box_utils": box_utils, "draw_umich_gaussian": cns["draw_umich_gaussian"],
           "gaussian_radius": cns["gaussian_radius"], "draw_seg_mask": cns["draw_seg_mask"]}
    extract("models/dense_heads/keypoint_assigner/centertarget_assigner.py", ["CenterTargetAssigner"], ans)
    assigner = ans["CenterTargetAssigner"](model_cfg=AD(TARGET_ASSIGNER_CONFIG=TARGET_CFG), voxel_size=VOXEL, point_cloud_range=RANGE, class_names=CLASSES)
    gt = gt_batch(rng)
    tg = assigner.assign_targets(torch.from_numpy(gt))
    keep = ("hm_target", "anno_box_target", "ind_target", "mask_target", "src_box_target", "xsys_target")
    np.savez_compressed(os.path.join(OUT, "pyref_center_targets.npz"), gt_boxes=gt, point_cloud_range=RANGE, voxel_size=np.array(VOXEL, np.float64),
                        **{k: tg[k].numpy() for k in keep})
    print("center_targets", {k: tuple(tg[k].shape) for k in keep}, "objects", int(tg["mask_target"].sum()))

    # ---- the head's losses on random head maps ------------------------------------------------------------------------------------------
    torch.cuda.FloatTensor = torch.FloatTensor
    lns = {"torch": torch, "nn": torch.nn, "F": torch.nn.functional, "np": np, "center_utils": center_utils, "box_utils": box_utils}
    loss_names = ["CenterNetFocalLoss", "CenterNetResLoss", "CenterNetRotBinResLoss", "SigmoidFocalClassificationLoss", "get_corner_loss_mse",
                  "get_iouscore_loss_bce", "get_rot_binres_loss"]
    extract("utils/loss_utils.py", loss_names, lns)
    loss_utils = types.SimpleNamespace(**{k: lns[k] for k in loss_names})
    iou3d = types.SimpleNamespace(boxes_iou3d_gpu=lambda a, b: torch.from_numpy(oracle.boxes_iou3d(a.detach().numpy(), b.detach().numpy())))
    hns = {"torch": torch, "np": np, "nn": torch.nn, "loss_utils": loss_utils, "box_utils": box_utils, "center_utils": center_utils,
           "iou3d_nms_utils": iou3d}
    rel = "models/dense_heads/center_af_head_template.py"

    class Stub(torch.nn.Module):
        pass
    for name in ("build_losses", "get_loss", "get_hm_loss", "get_offset_loss", "get_height_loss", "get_dim_loss", "get_rot_loss", "get_segm_loss",
                 "get_corner_loss", "get_iouscore_loss", "get_max_iou_with_same_class", "gthm_based_predicted_boxes_generation",
                 "predhm_based_predicted_boxes_generation_ssd"):
        setattr(Stub, name, method(rel, "CenterAFHeadTemplate", name, hns))
    head = Stub()
    head.model_cfg = AD(LOSS_CONFIG=LOSS_CFG)
    head.num_class, head.feature_map_stride, head.voxel_size, head.point_cloud_range = 3, 8, VOXEL, RANGE
    head.forward_ret_dict = {}
    head.build_losses(LOSS_CFG)
    b, (sy, sx) = gt.shape[0], tg["hm_target"].shape[2:]
    chans = (("hm", 3), ("offset", 2), ("height", 1), ("dim", 3), ("rot", 24), ("segm", 1), ("iouscore", 1))
    preds = {k: (torch.from_numpy(rng.standard_normal((b, c, sy, sx)).astype(np.float32)) * (2.0 if k == "hm" else 0.7)).requires_grad_(True) for k, c in chans}
    # decodable boxes near the ground truth, so that the IoU labels cover foreground, interval and background: at every object's pixel
    # the regression maps hold the target plus noise, the heat map a peak
    with torch.no_grad():
        for s in range(b):
            for k in range(tg["mask_target"].shape[1]):
                if not tg["mask_target"][s, k]:
                    continue
                ind = int(tg["ind_target"][s, k]); y, x = divmod(ind, sx)
                noise = [0.0, 0.05, 0.3][k % 3]
                preds["offset"][s, :, y, x] = tg["anno_box_target"][s, k, 0:2] + noise * torch.randn(2)
                preds["height"][s, 0, y, x] = tg["anno_box_target"][s, k, 2] + noise * float(torch.randn(()))
                preds["dim"][s, :, y, x] = tg["anno_box_target"][s, k, 3:6] * (1 + noise * torch.randn(3)).clamp(0.5, 1.5)
                ry = float(tg["anno_box_target"][s, k, 6]) % (2 * math.pi)
                per = 2 * math.pi / 12
                sh = (ry + per / 2) % (2 * math.pi)
                which = int(sh // per)
                preds["rot"][s, :12, y, x] = -3.0
                preds["rot"][s, which, y, x] = 3.0
                preds["rot"][s, 12 + which, y, x] = (sh - (which * per + per / 2)) / (per / 2) + noise * float(torch.randn(()))
                preds["hm"][s, int(gt[s, k, 7]) - 1, y, x] = 6.0 + k * 0.01
    segm_t = (torch.from_numpy(rng.uniform(0, 1, (b, 1, sy, sx))) < 0.15).float()
    segm_t[1] = 0            # a sample without a foreground pixel: its normaliser clamps at 1
    head.forward_ret_dict.update({k + "_pred": v for k, v in preds.items()})
    head.forward_ret_dict.update(tg)
    head.forward_ret_dict["segm_target"] = segm_t
    head.forward_ret_dict.update(head.predhm_based_predicted_boxes_generation_ssd(K=24))
    head.forward_ret_dict.update(head.gthm_based_predicted_boxes_generation())
    loss, tb = head.get_loss()
    grads = torch.autograd.grad(loss, list(preds.values()))
    out = {"pred_" + k: v.detach().numpy() for k, v in preds.items()}
    out.update({"grad_" + k: g.numpy() for k, g in zip(preds, grads)})
    out.update({"term_" + k.replace("rpn_", "").replace("_loss", ""): np.float64(v) for k, v in tb.items() if k.endswith("_loss") and k != "rpn_loss"})
    out.update(loss=np.float64(loss.item()), segm_target=segm_t.numpy(), gt_boxes=gt, point_cloud_range=RANGE, voxel_size=np.array(VOXEL, np.float64),
               decoded_topk_boxes=head.forward_ret_dict["batch_box_preds"].detach().numpy(), decoded_gt_boxes=head.forward_ret_dict["gthm_box_preds"].detach().numpy(),
               num_fg=np.float64(tb["num_sample_fg"]), num_bg=np.float64(tb["num_sample_bg"]), num_inter=np.float64(tb["num_sample_inter"]))
    out.update({k: tg[k].numpy() for k in keep})
    np.savez_compressed(os.path.join(OUT, "pyref_center_losses.npz"), **out)
    print("center_losses", float(loss), {k: round(float(v), 5) for k, v in tb.items() if "loss" in k},
          "fg/bg/inter per sample", float(tb["num_sample_fg"]), float(tb["num_sample_bg"]), float(tb["num_sample_inter"]))


if __name__ == "__main__":
    main()
