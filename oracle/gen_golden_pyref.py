"""Generates tests/golden/pyref_*.npz from the reference's own pure-Python functions (build container only).

Pins, with outputs of the reference code itself, the pieces of the hot path and of the replay harness that the reference
implements in importable Python:

  spconv/structure.py            scatter_nd, SparseConvTensor.dense()                      (imported by path: numpy + torch only)
  spconv/ops.py:20-43            get_conv_output_size, get_deconv_output_size              (function defs compiled out of the file)
  vfe/mean_vfe.py:14-31          MeanVFE.forward                                            (method compiled out of the class)
  utils/common_utils.py          get_voxel_centers, rotate_points_along_z, limit_period     (imported by path)
  utils/box_utils.py             boxes_to_corners_3d, in_hull, boxes_iou_normal,
                                 boxes3d_lidar_to_aligned_bev_boxes, boxes3d_nearest_bev_iou, enlarge_box3d
  utils/box_coder_utils.py       ResidualCoder.encode_torch / decode_torch                  (imported by path)
  utils/loss_utils.py            SigmoidFocalClassificationLoss, WeightedSmoothL1Loss.smooth_l1_loss
  utils/loss_utils.py:217-241    get_corner_loss_lidar
  roi_heads/roi_withiou_head_template.py:133-195   RoIWithIoUHeadTemplate.get_box_reg_layer_loss (stub head, fv2p.yaml LOSS_CONFIG)
  roi_heads/target_assigner/proposal_target_layer.py:20-86   ProposalTargetLayer.forward (labels from the overlaps; stub sampler)
  roi_heads/roi_withiou_head_template.py:196-265   get_box_cls_layer_loss, get_box_iouscore_layer_loss
  roi_heads/iouguided_roi_head.py:195-220   get_global_grid_points_of_roi / get_dense_grid_points (methods compiled out of the class)
  backbones_2d/base_bev_backbone.py   BaseBEVBackbone (imported by path)
  dense_heads/point_head_simple.py:21-50, point_head_template.py:49-164   assign_targets, assign_stack_targets, get_cls_layer_loss
                                 (stub head; the point-in-box op answered by oracle.points_in_boxes_gpu)
  dense_heads/anchor_head_template.py:101-218   get_cls_layer_loss, get_box_reg_layer_loss, add_sin_difference, get_direction_target (stub head)
  dense_heads/anchor_head_template.py:229-276   AnchorHeadTemplate.generate_predicted_boxes (stub head)
  dense_heads/target_assigner/anchor_generator.py:17-61   AnchorGenerator.generate_anchors (imported by path)
  roi_heads/roi_withiou_head_template.py:101-131   RoIWithIoUHeadTemplate.assign_targets (canonical transformation; stub target layer)
  backbones_3d/pfe/residual_v2p_decoder.py:46-313 + pointnet2_batch/pointnet2_utils.py:292-326   ResidualVoxelToPointDecoder around top3_interpolate
  pointnet2_batch/pointnet2_modules.py:10-98 + pointnet2_utils.py:231-264   PointnetSAModuleMSG around QueryAndGroup (grid set abstraction)
  roi_heads/iouguided_roi_head.py:144-193 + roipoint_pool3d/roipoint_pool3d_utils.py:31-63   roipool3d_gpu around RoIPointPool3dFunction
  roi_heads/iouguided_roi_head.py:11-304   IoUGuidedRoIHead.__init__ + forward_single_loop over the classes above (whole second-stage forward)
  roi_heads/roi_head_template.py:46-99 + model_utils/model_nms_utils.py:6-25   RoIHeadTemplate.proposal_layer around class_agnostic_nms
  datasets/kitti/kitti_object_eval_python/rotate_iou.py:17-260   inter, devRotateIoUEval (numba.cuda device code run as plain Python)
  datasets/kitti/kitti_object_eval_python/eval.py:121-147        d3_box_overlap_kernel (3-D IoU from the BEV intersection; def compiled out of the file)
  ops/spconv/test_utils.py:144-193   generate_sparse_data (the reference's sparse / dense test-data generator)
  dense_heads/target_assigner/axis_aligned_target_assigner.py:130-215   AxisAlignedTargetAssigner.assign_targets_single
                                 (method compiled out of the class; Tensor.cuda is the identity while it runs)

Modules that import the CUDA extensions at load time cannot be imported as a whole, so single definitions are taken out
of their syntax tree and compiled against torch (CPU).  Nothing of the reference is copied into the repository: the
fixtures hold inputs and outputs only.  Run:   python oracle/gen_golden_pyref.py
"""
import ast
import importlib.util
import os

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/pcdet"
OUT = os.path.join(REPO, "tests", "golden")


def by_path(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def extract(rel, names, ns):
    """Compiles the named top-level defs / classes of a reference file into the namespace `ns`."""
    path = os.path.join(REF, rel)
    tree = ast.parse(open(path).read())
    body = [n for n in tree.body if isinstance(n, (ast.FunctionDef, ast.ClassDef)) and n.name in names]
    assert len(body) == len(names), (rel, names)
    exec(compile(ast.Module(body=body, type_ignores=[]), path, "exec"), ns)
    return ns


def method(rel, cls, name, ns):
    path = os.path.join(REF, rel)
    tree = ast.parse(open(path).read())
    c = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == cls][0]
    f = [n for n in c.body if isinstance(n, ast.FunctionDef) and n.name == name]
    assert len(f) == 1
    exec(compile(ast.Module(body=f, type_ignores=[]), path, "exec"), ns)
    return ns[name]


def random_boxes(rng, n, spread=20.0):
    xy = rng.uniform(-spread, spread, size=(n, 2))
    z = rng.uniform(-1.5, 0.5, size=(n, 1))
    dims = np.array([3.9, 1.6, 1.56]) * rng.uniform(0.7, 1.3, size=(n, 3))
    yaw = rng.uniform(-np.pi, np.pi, size=(n, 1))
    return np.concatenate([xy, z, dims, yaw], 1).astype(np.float32)


def main():
    import sys
    only = set(sys.argv[1:])       # `python oracle/gen_golden_pyref.py rotate_iou` rewrites that fixture only
    save = lambda name, **kw: (not only or name in only) and (np.savez_compressed(os.path.join(OUT, f"pyref_{name}.npz"), **kw), print(name, {k: np.asarray(v).shape for k, v in kw.items()}))
    rng = np.random.default_rng(2024)

    # ---- scatter_nd / dense ------------------------------------------------------------------------------------------------
    st = by_path("ref_structure", "ops/spconv/structure.py")
    shape, batch, c = [5, 12, 9], 3, 7
    cells = rng.permutation(batch * int(np.prod(shape)))[:200]
    idx = np.stack(np.unravel_index(cells, [batch] + shape), 1).astype(np.int32)
    feats = rng.standard_normal((200, c)).astype(np.float32)
    x = st.SparseConvTensor(torch.from_numpy(feats), torch.from_numpy(idx), shape, batch)
    save("dense", features=feats, indices=idx, spatial_shape=np.array(shape), batch_size=batch,
         dense_channels_first=x.dense().numpy(), dense_channels_last=x.dense(channels_first=False).numpy(),
         scatter_nd=st.scatter_nd(torch.from_numpy(idx).long(), torch.from_numpy(feats), [batch] + shape + [c]).numpy())

    # ---- conv output sizes ---------------------------------------------------------------------------------------------------
    ns = extract("ops/spconv/ops.py", ["get_conv_output_size", "get_deconv_output_size"], {})
    cases, conv, deconv = [], [], []
    for size in ([41, 1600, 1408], [21, 800, 704], [11, 400, 352], [5, 200, 176], [7, 9, 10], [40, 1504, 1504]):
        for k, s, p, d in (([3, 3, 3], [2, 2, 2], [1, 1, 1], [1, 1, 1]), ([3, 3, 3], [2, 2, 2], [0, 1, 1], [1, 1, 1]),
                           ([3, 1, 1], [2, 1, 1], [0, 0, 0], [1, 1, 1]), ([3, 3, 3], [1, 1, 1], [1, 1, 1], [1, 1, 1]),
                           ([2, 2, 2], [2, 2, 2], [0, 0, 0], [1, 1, 1]), ([3, 3, 3], [1, 1, 1], [2, 2, 2], [2, 2, 2])):
            cases.append(size + k + s + p + d)
            conv.append(ns["get_conv_output_size"](size, k, s, p, d))
            deconv.append(ns["get_deconv_output_size"](size, k, s, p, d, [0, 0, 0]))
    save("conv_output_size", cases=np.array(cases), conv=np.array(conv), deconv=np.array(deconv))

    # ---- MeanVFE -----------------------------------------------------------------------------------------------------------------
    fwd = method("models/backbones_3d/vfe/mean_vfe.py", "MeanVFE", "forward", {"torch": torch})
    num = rng.integers(0, 6, size=300).astype(np.int32)      # 0 .. 5 points, zero-point voxels included
    vox = rng.standard_normal((300, 5, 4)).astype(np.float32)
    vox *= (np.arange(5)[None, :, None] < num[:, None, None])   # zero padding past num_points, as the voxeliser writes it
    bd = {"voxels": torch.from_numpy(vox), "voxel_num_points": torch.from_numpy(num)}
    save("mean_vfe", voxels=vox, num_points=num, voxel_features=fwd(None, bd)["voxel_features"].numpy())

    # ---- common_utils --------------------------------------------------------------------------------------------------------------
    cu = by_path("ref_common_utils", "utils/common_utils.py")
    coords = np.stack([rng.integers(0, 41, 400), rng.integers(0, 1600, 400), rng.integers(0, 1408, 400)], 1).astype(np.int32)
    centres = {str(f): cu.get_voxel_centers(torch.from_numpy(coords), f, [0.05, 0.05, 0.1], [0, -40, -3, 70.4, 40, 1]).numpy() for f in (1, 2, 4, 8)}
    pts = rng.standard_normal((50, 6, 5)).astype(np.float32)
    ang = rng.uniform(-4, 4, 50).astype(np.float32)
    val = rng.uniform(-10, 10, 200).astype(np.float32)
    save("common_utils", coords=coords, **{f"centres_x{k}": v for k, v in centres.items()}, points=pts, angle=ang,
         rotated=cu.rotate_points_along_z(torch.from_numpy(pts), torch.from_numpy(ang)).numpy(), val=val,
         limit_period_pi=cu.limit_period(torch.from_numpy(val), 0.5, np.pi).numpy(),
         limit_period_2pi=cu.limit_period(torch.from_numpy(val), 0.0, 2 * np.pi).numpy())

    # ---- box_utils ---------------------------------------------------------------------------------------------------------------------
    import scipy
    from scipy.spatial import Delaunay
    bns = {"torch": torch, "np": np, "scipy": scipy, "Delaunay": Delaunay, "common_utils": cu}
    extract("utils/box_utils.py", ["in_hull", "boxes_to_corners_3d", "boxes_iou_normal", "boxes3d_lidar_to_aligned_bev_boxes",
                                   "boxes3d_nearest_bev_iou", "enlarge_box3d"], bns)
    boxes = random_boxes(rng, 24, spread=8.0)
    corners = bns["boxes_to_corners_3d"](torch.from_numpy(boxes)).numpy()
    pts = rng.uniform(-11, 11, size=(4000, 3)).astype(np.float32)
    pts[:, 2] = rng.uniform(-2.5, 1.5, 4000)
    # keep points at least 2 cm away from every face of every box: the hull test, the CUDA point-in-box test (margin 1e-5) and the CPU one (margin 1e-2) all agree there
    keep = np.ones(len(pts), bool)
    for b in boxes:
        loc = pts - b[:3]
        c, s = np.cos(-b[6]), np.sin(-b[6])
        lx, ly = loc[:, 0] * c - loc[:, 1] * s, loc[:, 0] * s + loc[:, 1] * c
        for v, h in ((lx, b[3] / 2), (ly, b[4] / 2), (loc[:, 2], b[5] / 2)):
            keep &= np.abs(np.abs(v) - h) > 2e-2
    pts = pts[keep]
    inside = np.stack([bns["in_hull"](pts.astype(np.float64), corners[i].astype(np.float64)) for i in range(len(boxes))])
    a, b2 = random_boxes(rng, 60, 6.0), random_boxes(rng, 40, 6.0)
    bev_a = bns["boxes3d_lidar_to_aligned_bev_boxes"](torch.from_numpy(a))
    save("box_utils", boxes=boxes, corners=corners, points=pts, inside=inside.astype(np.int32), boxes_a=a, boxes_b=b2,
         aligned_bev_a=bev_a.numpy(), nearest_bev_iou=bns["boxes3d_nearest_bev_iou"](torch.from_numpy(a), torch.from_numpy(b2)).numpy(),
         iou_normal_aa=bns["boxes_iou_normal"](bev_a, bev_a).numpy(),
         enlarged=bns["enlarge_box3d"](torch.from_numpy(boxes), [0.2, 0.2, 0.2]).numpy())

    # ---- box coder + losses ------------------------------------------------------------------------------------------------------------
    bc = by_path("ref_box_coder", "utils/box_coder_utils.py")
    coder = bc.ResidualCoder()
    g, an = random_boxes(rng, 100), random_boxes(rng, 100)
    enc = coder.encode_torch(torch.from_numpy(g.copy()), torch.from_numpy(an.copy()))
    dec = coder.decode_torch(enc, torch.from_numpy(an.copy()))
    lns = {"torch": torch, "np": np, "nn": torch.nn, "F": torch.nn.functional}
    extract("utils/loss_utils.py", ["SigmoidFocalClassificationLoss"], lns)
    sl1 = method("utils/loss_utils.py", "WeightedSmoothL1Loss", "smooth_l1_loss", {"torch": torch})
    logits = rng.standard_normal((2, 500, 1)).astype(np.float32) * 3
    onehot = (rng.uniform(size=(2, 500, 1)) < 0.1).astype(np.float32)
    w = rng.uniform(0, 1, size=(2, 500)).astype(np.float32)
    focal = lns["SigmoidFocalClassificationLoss"](alpha=0.25, gamma=2.0)(torch.from_numpy(logits), torch.from_numpy(onehot), torch.from_numpy(w))
    diff = rng.standard_normal(1000).astype(np.float32)
    save("coder_losses", gt=g, anchors=an, encoded=enc.numpy(), decoded=dec.numpy(), logits=logits, onehot=onehot, weights=w,
         focal=focal.numpy(), diff=diff, smooth_l1_beta9=sl1(torch.from_numpy(diff), 1.0 / 9.0).numpy(),
         smooth_l1_beta1=sl1(torch.from_numpy(diff), 1.0).numpy())

    # ---- corner loss --------------------------------------------------------------------------------------------------------------------
    cns = {"torch": torch, "np": np, "nn": torch.nn, "F": torch.nn.functional, "box_utils": None}
    import types as _types
    cns["box_utils"] = _types.SimpleNamespace(boxes_to_corners_3d=bns["boxes_to_corners_3d"])
    extract("utils/loss_utils.py", ["WeightedSmoothL1Loss", "get_corner_loss_lidar"], cns)
    pb, gb = random_boxes(rng, 200, 10.0), random_boxes(rng, 200, 10.0)
    gb[:100] = pb[:100] + rng.normal(0, 0.15, size=(100, 7)).astype(np.float32)      # near pairs: the quadratic branch of smooth-L1
    gb[50:100, 6] += np.pi                                                            # ... and the heading-flipped twin
    save("corner_loss", pred=pb, gt=gb, loss=cns["get_corner_loss_lidar"](torch.from_numpy(pb), torch.from_numpy(gb)).numpy())

    # ---- second-stage canonical transformation ----------------------------------------------------------------------------------------
    # RoIWithIoUHeadTemplate.assign_targets (roi_withiou_head_template.py:101-131) behind a stub proposal_target_layer that hands
    # over the sampled rois / boxes: what remains is the roi-frame transformation and the heading folding
    canon = method("models/roi_heads/roi_withiou_head_template.py", "RoIWithIoUHeadTemplate", "assign_targets", {"torch": torch, "np": np, "common_utils": cu})
    rois = np.stack([random_boxes(rng, 64, 15.0), random_boxes(rng, 64, 15.0)])
    gts = rois + rng.normal(0, 0.4, size=rois.shape).astype(np.float32)
    gts[..., 6] = rng.uniform(-2 * np.pi, 2 * np.pi, size=gts.shape[:2])          # every heading quadrant
    gts = np.concatenate([gts, np.ones(gts.shape[:2] + (1,), np.float32)], -1).astype(np.float32)
    stub = _types.SimpleNamespace(proposal_target_layer=_types.SimpleNamespace(
        forward=lambda bd: {"rois": torch.from_numpy(rois.copy()), "gt_of_rois": torch.from_numpy(gts.copy())}))
    td = canon(stub, {"batch_size": 2})
    save("canonical_targets", rois=rois, gt_of_rois=gts, canonical=td["gt_of_rois"].numpy(), src=td["gt_of_rois_src"].numpy())

    # ---- second-stage regression + corner loss ------------------------------------------------------------------------------------------
    # RoIWithIoUHeadTemplate.get_box_reg_layer_loss (roi_withiou_head_template.py:133-195) behind a stub head with fv2p.yaml's
    # LOSS_CONFIG (smooth-l1, CORNER_LOSS_REGULARIZATION, unit weights); inputs: the canonical targets made above
    reg_loss = method("models/roi_heads/roi_withiou_head_template.py", "RoIWithIoUHeadTemplate", "get_box_reg_layer_loss",
                      {"torch": torch, "np": np, "F": torch.nn.functional, "common_utils": cu,
                       "loss_utils": _types.SimpleNamespace(get_corner_loss_lidar=cns["get_corner_loss_lidar"])})
    keep_cuda2 = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        l1 = cns["WeightedSmoothL1Loss"](code_weights=[1.0] * 7)
    finally:
        torch.Tensor.cuda = keep_cuda2
    rstub = _types.SimpleNamespace(box_coder=coder, reg_loss_func=l1, model_cfg=_types.SimpleNamespace(LOSS_CONFIG=_types.SimpleNamespace(
        REG_LOSS="smooth-l1", CORNER_LOSS_REGULARIZATION=True, LOSS_WEIGHTS={"rcnn_reg_weight": 1.0, "rcnn_corner_weight": 1.0})))
    rcnn_reg = rng.normal(0, 0.3, size=(128, 7)).astype(np.float32)
    valid = (rng.uniform(size=(2, 64)) < 0.4).astype(np.int64)
    frd = {"reg_valid_mask": torch.from_numpy(valid), "gt_of_rois": td["gt_of_rois"].clone(), "gt_of_rois_src": td["gt_of_rois_src"].clone(),
           "rcnn_reg": torch.from_numpy(rcnn_reg), "rois": torch.from_numpy(rois.copy())}
    total, tb = reg_loss(rstub, frd)
    save("roi_reg_loss", rois=rois, gt_canonical=td["gt_of_rois"].numpy(), gt_src=td["gt_of_rois_src"].numpy(), rcnn_reg=rcnn_reg, valid=valid,
         loss_reg=np.float32(tb["rcnn_loss_reg"]), loss_corner=np.float32(tb["rcnn_loss_corner"]), total=np.float32(total.item()))

    # ---- second-stage labels, classification and IoU-score losses, total ----------------------------------------------------------------
    # ProposalTargetLayer.forward (proposal_target_layer.py:20-86, CLS_SCORE_TYPE roi_iou) behind a stub sampler, then
    # get_box_cls_layer_loss / get_box_iouscore_layer_loss / get_box_reg_layer_loss (roi_withiou_head_template.py:133-265)
    ptl_fwd = method("models/roi_heads/target_assigner/proposal_target_layer.py", "ProposalTargetLayer", "forward", {"torch": torch, "np": np})
    ious = rng.uniform(0, 1, size=(2, 64)).astype(np.float32)
    ious[0, :4] = [0.25, 0.75, 0.55, 0.5500001]                                       # the thresholds themselves
    tcfg = _types.SimpleNamespace(REG_FG_THRESH=0.55, CLS_SCORE_TYPE="roi_iou", CLS_BG_THRESH=0.25, CLS_FG_THRESH=0.75)
    ptl = _types.SimpleNamespace(roi_sampler_cfg=tcfg, sample_rois_for_rcnn=lambda batch_dict: (
        torch.from_numpy(rois.copy()), torch.from_numpy(gts.copy()), torch.from_numpy(ious.copy()), torch.zeros(2, 64), torch.ones(2, 64, dtype=torch.long)))
    tgt = ptl_fwd(ptl, {"batch_size": 2})
    lns2 = {"torch": torch, "np": np, "F": torch.nn.functional, "common_utils": cu,
            "loss_utils": _types.SimpleNamespace(get_corner_loss_lidar=cns["get_corner_loss_lidar"])}
    cls_loss = method("models/roi_heads/roi_withiou_head_template.py", "RoIWithIoUHeadTemplate", "get_box_cls_layer_loss", dict(lns2))
    iou_loss = method("models/roi_heads/roi_withiou_head_template.py", "RoIWithIoUHeadTemplate", "get_box_iouscore_layer_loss", dict(lns2))
    hstub = _types.SimpleNamespace(box_coder=coder, reg_loss_func=l1, model_cfg=_types.SimpleNamespace(
        TARGET_CONFIG=tcfg, LOSS_CONFIG=_types.SimpleNamespace(
            REG_LOSS="smooth-l1", CLS_LOSS="BinaryCrossEntropy", IOUSCORE_LOSS="smooth-l1", CORNER_LOSS_REGULARIZATION=True,
            LOSS_WEIGHTS={"rcnn_cls_weight": 1.0, "rcnn_reg_weight": 1.0, "rcnn_corner_weight": 1.0, "rcnn_iouscore_weight": 1.0})))
    rcnn_cls = rng.normal(0, 1.5, size=(128, 1)).astype(np.float32)
    rcnn_iou = rng.normal(0, 0.5, size=(128, 1)).astype(np.float32)
    frd2 = {"rcnn_cls": torch.from_numpy(rcnn_cls), "rcnn_cls_labels": tgt["rcnn_cls_labels"], "distribution_dict": {}, "rcnn_iouscore": torch.from_numpy(rcnn_iou),
            "batch_size": 2, "gt_iou_of_rois": tgt["gt_iou_of_rois"], "reg_valid_mask": tgt["reg_valid_mask"], "gt_of_rois": td["gt_of_rois"].clone(),
            "gt_of_rois_src": td["gt_of_rois_src"].clone(), "rcnn_reg": torch.from_numpy(rcnn_reg), "rois": torch.from_numpy(rois.copy())}
    l_cls, _ = cls_loss(hstub, frd2)
    l_iou, _ = iou_loss(hstub, frd2)
    l_reg, _ = reg_loss(hstub, frd2)
    save("roi_losses", rois=rois, gt_canonical=td["gt_of_rois"].numpy(), gt_src=td["gt_of_rois_src"].numpy(), ious=ious, rcnn_cls=rcnn_cls, rcnn_reg=rcnn_reg,
         rcnn_iou=rcnn_iou, cls_labels=tgt["rcnn_cls_labels"].numpy(), reg_valid=tgt["reg_valid_mask"].numpy(), loss_cls=np.float32(l_cls.item()),
         loss_iou=np.float32(float(l_iou)), loss_reg=np.float32(l_reg.item()), total=np.float32(l_cls.item() + float(l_iou) + l_reg.item()))

    # ---- RoI grid points ------------------------------------------------------------------------------------------------------------------
    gns = {"torch": torch, "np": np, "common_utils": cu}
    dense_pts = method("models/roi_heads/iouguided_roi_head.py", "IoUGuidedRoIHead", "get_dense_grid_points", dict(gns))
    glob_pts = method("models/roi_heads/iouguided_roi_head.py", "IoUGuidedRoIHead", "get_global_grid_points_of_roi", dict(gns))
    rb = np.stack([random_boxes(rng, 40, 20.0), random_boxes(rng, 40, 20.0)])
    world, local = glob_pts(_types.SimpleNamespace(get_dense_grid_points=dense_pts), torch.from_numpy(rb), 6)
    save("roi_grid_points", rois=rb, world=world.numpy(), local=local.numpy())

    # ---- first-stage box decoding ----------------------------------------------------------------------------------------------------------
    # AnchorHeadTemplate.generate_predicted_boxes (anchor_head_template.py:229-276) behind a stub head: ResidualCoder decoding against the
    # anchors and the direction-bin correction of the heading (DIR_OFFSET 0.78539, DIR_LIMIT_OFFSET 0, two bins)
    gen_boxes = method("models/dense_heads/anchor_head_template.py", "AnchorHeadTemplate", "generate_predicted_boxes",
                       {"torch": torch, "np": np, "common_utils": cu, "box_coder_utils": bc})
    anc = torch.from_numpy(random_boxes(rng, 300, 30.0))
    box_p = torch.from_numpy(rng.normal(0, 0.4, size=(2, 300, 7)).astype(np.float32))
    dir_p = torch.from_numpy(rng.normal(0, 1.0, size=(2, 300, 2)).astype(np.float32))
    cls_p = torch.from_numpy(rng.normal(0, 1.0, size=(2, 300, 1)).astype(np.float32))
    head_stub = _types.SimpleNamespace(anchors=anc, use_multihead=False, box_coder=coder,
                                       model_cfg=_types.SimpleNamespace(DIR_OFFSET=0.78539, DIR_LIMIT_OFFSET=0.0, NUM_DIR_BINS=2))
    _, dec = gen_boxes(head_stub, 2, cls_p, box_p.clone(), dir_p)
    save("predicted_boxes", anchors=anc.numpy(), box_preds=box_p.numpy(), dir_preds=dir_p.numpy(), boxes=dec.numpy())

    # ---- anchors -------------------------------------------------------------------------------------------------------------------------
    # AnchorGenerator.generate_anchors (anchor_generator.py:17-61; `.cuda()` an identity here) with fv2p.yaml's ANCHOR_GENERATOR_CONFIG
    # (Car: 3.9 x 1.6 x 1.56, rotations 0 / 1.57, bottom -1.78, align_center False) on the KITTI BEV map and on the reduced map
    # of the parity tests; flattened the way AnchorHeadTemplate uses them (y, x, rotation)
    import hashlib
    ag = by_path("ref_anchor_generator", "models/dense_heads/target_assigner/anchor_generator.py")
    acfg = [{"anchor_sizes": [[3.9, 1.6, 1.56]], "anchor_rotations": [0, 1.57], "anchor_bottom_heights": [-1.78], "align_center": False}]
    keep_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        big = ag.AnchorGenerator([0, -40.0, -3, 70.4, 40.0, 1], acfg).generate_anchors([np.array([176, 200])])[0][0].view(-1, 7).numpy()
        small = ag.AnchorGenerator([0, -20.0, -3, 35.2, 20.0, 1], acfg).generate_anchors([np.array([88, 100])])[0][0].view(-1, 7).numpy()
    finally:
        torch.Tensor.cuda = keep_cuda
    save("anchors", small=small, kitti_every_97th=big[::97].copy(), kitti_shape=np.array(big.shape),
         kitti_sha256=np.frombuffer(hashlib.sha256(np.ascontiguousarray(big).tobytes()).digest(), np.uint8))

    # ---- first-stage target assignment ------------------------------------------------------------------------------------------------
    # AxisAlignedTargetAssigner.assign_targets_single (axis_aligned_target_assigner.py:130-215) as the reference runs it for
    # fv2p.yaml (match_height False, POS_FRACTION -1, NORM_BY_NUM_EXAMPLES False).  The method moves index tensors with
    # `.cuda()`; the container has no GPU, so Tensor.cuda is the identity while it runs (a runtime shim, the source is untouched).
    import types
    sys_path_root = os.path.dirname(REPO)
    import sys
    for pth in (REPO, os.path.join(REPO, "from-voxel-to-point_amd")):
        if pth not in sys.path:
            sys.path.insert(0, pth)
    from fv2p_harness import synth
    from fv2p_harness.fv2p_model import AnchorHead, FV2PConfig

    class Small(FV2PConfig):   # tests/test_fv2p_step_gpu.py::SmallFV2P, the Car entry of ANCHOR_GENERATOR_CONFIG only
        point_cloud_range = (0.0, -20.0, -3.0, 35.2, 20.0, 1.0)
        grid_size = (704, 800, 40)
        anchor_classes = FV2PConfig.anchor_classes[:1]
    anchors = AnchorHead(Small, 128).anchors.clone()
    assign = method("models/dense_heads/target_assigner/axis_aligned_target_assigner.py", "AxisAlignedTargetAssigner", "assign_targets_single",
                    {"torch": torch, "np": np, "box_utils": types.SimpleNamespace(boxes3d_nearest_bev_iou=bns["boxes3d_nearest_bev_iou"]),
                     "iou3d_nms_utils": None})
    me = types.SimpleNamespace(match_height=False, pos_fraction=None, sample_size=512, norm_by_num_examples=False, box_coder=coder)
    keep_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        out = {}
        for smp in range(2):
            _, bx = synth.lidar_cloud(31 + smp, 2048, pc_range=np.array(Small.point_cloud_range, np.float32), return_boxes=True)
            if smp == 1:   # a box outside every anchor: its maximum overlap is zero, it must force nothing
                bx = np.concatenate([bx, np.array([[500.0, 500.0, 0.0, 3.9, 1.6, 1.5, 0.3]], np.float32)])
            gt = torch.from_numpy(bx.astype(np.float32))
            r = assign(me, anchors, gt, torch.ones(len(bx), dtype=torch.int32), matched_threshold=Small.anchor_classes[0][2], unmatched_threshold=Small.anchor_classes[0][3])
            out[f"gt{smp}"] = bx.astype(np.float32)
            out[f"labels{smp}"] = r["box_cls_labels"].numpy()
            out[f"targets{smp}"] = r["box_reg_targets"].numpy()
    finally:
        torch.Tensor.cuda = keep_cuda
    save("anchor_assign", anchors=anchors.numpy(), **out)

    # ---- the three anchor sets of fv2p.yaml and the batch-level assignment -----------------------------------------------------------------
    # ANCHOR_GENERATOR_CONFIG has a Car, a Pedestrian and a Cyclist entry although CLASS_NAMES is ['Car']: AnchorGenerator builds all
    # three sets, AnchorHeadTemplate concatenates them (torch.cat(anchors, dim=-3)), and AxisAlignedTargetAssigner.assign_targets
    # (:36-128) assigns class by class — the Pedestrian / Cyclist anchors never meet a box of their class
    acfg3 = [dict(class_name="Car", anchor_sizes=[[3.9, 1.6, 1.56]], anchor_rotations=[0, 1.57], anchor_bottom_heights=[-1.78], align_center=False,
                  matched_threshold=0.6, unmatched_threshold=0.45),
             dict(class_name="Pedestrian", anchor_sizes=[[0.8, 0.6, 1.73]], anchor_rotations=[0, 1.57], anchor_bottom_heights=[-0.6], align_center=False,
                  matched_threshold=0.5, unmatched_threshold=0.35),
             dict(class_name="Cyclist", anchor_sizes=[[1.76, 0.6, 1.73]], anchor_rotations=[0, 1.57], anchor_bottom_heights=[-0.6], align_center=False,
                  matched_threshold=0.5, unmatched_threshold=0.35)]
    batch_assign = method("models/dense_heads/target_assigner/axis_aligned_target_assigner.py", "AxisAlignedTargetAssigner", "assign_targets",
                          {"torch": torch, "np": np})
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        sets3 = ag.AnchorGenerator([0, -20.0, -3, 35.2, 20.0, 1], acfg3).generate_anchors([np.array([88, 100])] * 3)[0]
        flat3 = torch.cat(sets3, dim=-3).view(-1, 7)
        me3 = types.SimpleNamespace(match_height=False, pos_fraction=None, sample_size=512, norm_by_num_examples=False, box_coder=coder,
                                    class_names=np.array(["Car"]), anchor_class_names=[c["class_name"] for c in acfg3], use_multihead=False,
                                    matched_thresholds={c["class_name"]: c["matched_threshold"] for c in acfg3},
                                    unmatched_thresholds={c["class_name"]: c["unmatched_threshold"] for c in acfg3})
        me3.assign_targets_single = lambda *a, **k: assign(me3, *a, **k)
        gt3 = np.zeros((2, 48, 8), np.float32)
        for smp in range(2):
            n = len(out[f"gt{smp}"])
            gt3[smp, :n, :7] = out[f"gt{smp}"]
            gt3[smp, :n, 7] = 1.0
        r3 = batch_assign(me3, sets3, torch.from_numpy(gt3))
    finally:
        torch.Tensor.cuda = keep_cuda
    save("anchor_assign3", anchors=flat3.numpy(), gt=gt3, labels=r3["box_cls_labels"].numpy(), targets=r3["box_reg_targets"].numpy())

    # ---- first-stage losses ---------------------------------------------------------------------------------------------------------------
    # AnchorHeadTemplate.get_cls_layer_loss + get_box_reg_layer_loss (anchor_head_template.py:101-218) behind a stub head with
    # fv2p.yaml's loss configuration, on the labels / targets the reference assigner produced above (every non-background anchor
    # of the two samples plus 2000 background ones)
    lab = np.stack([out["labels0"], out["labels1"]])
    tgt_reg = np.stack([out["targets0"], out["targets1"]])
    pick = np.unique(np.concatenate([np.nonzero((lab != 0).any(0))[0], rng.choice(lab.shape[1], 2000, replace=False)]))
    lab_s, tgt_s, anc_s = lab[:, pick].copy(), tgt_reg[:, pick].copy(), anchors.numpy()[pick].copy()
    n_s = len(pick)
    p_cls = rng.normal(-2.0, 2.0, size=(2, n_s, 1)).astype(np.float32)
    p_box = (tgt_s + rng.normal(0, 0.3, size=tgt_s.shape)).astype(np.float32)
    p_dir = rng.normal(0, 1.0, size=(2, n_s, 2)).astype(np.float32)
    ans = {"torch": torch, "np": np, "common_utils": cu, "box_coder_utils": bc}
    extract("utils/loss_utils.py", ["WeightedCrossEntropyLoss"], lns)
    sin_diff = method("models/dense_heads/anchor_head_template.py", "AnchorHeadTemplate", "add_sin_difference", dict(ans))
    dir_tgt = method("models/dense_heads/anchor_head_template.py", "AnchorHeadTemplate", "get_direction_target", dict(ans))
    a_cls = method("models/dense_heads/anchor_head_template.py", "AnchorHeadTemplate", "get_cls_layer_loss", dict(ans))
    a_reg = method("models/dense_heads/anchor_head_template.py", "AnchorHeadTemplate", "get_box_reg_layer_loss", dict(ans))
    astub = _types.SimpleNamespace(
        num_class=1, use_multihead=False, num_anchors_per_location=1, anchors=torch.from_numpy(anc_s), add_sin_difference=sin_diff, get_direction_target=dir_tgt,
        cls_loss_func=lns["SigmoidFocalClassificationLoss"](alpha=0.25, gamma=2.0), reg_loss_func=l1, dir_loss_func=lns["WeightedCrossEntropyLoss"](),
        model_cfg=_types.SimpleNamespace(DIR_OFFSET=0.78539, NUM_DIR_BINS=2, LOSS_CONFIG=_types.SimpleNamespace(
            LOSS_WEIGHTS={"cls_weight": 1.0, "loc_weight": 2.0, "dir_weight": 0.2})),
        forward_ret_dict={"cls_preds": torch.from_numpy(p_cls), "box_cls_labels": torch.from_numpy(lab_s.copy()), "box_preds": torch.from_numpy(p_box),
                          "dir_cls_preds": torch.from_numpy(p_dir), "box_reg_targets": torch.from_numpy(tgt_s)})
    lc, tbc = a_cls(astub)
    lb, tbb = a_reg(astub)
    save("anchor_losses", anchors=anc_s, labels=lab_s, reg_targets=tgt_s, cls=p_cls, box=p_box, dirs=p_dir, loss_cls=np.float32(tbc["rpn_loss_cls"]),
         loss_loc=np.float32(tbb["rpn_loss_loc"]), loss_dir=np.float32(tbb["rpn_loss_dir"]), total=np.float32(lc.item() + lb.item()))

    # ---- point head: targets and loss ------------------------------------------------------------------------------------------------------
    # PointHeadSimple.assign_targets + PointHeadTemplate.assign_stack_targets / get_cls_layer_loss (point_head_simple.py:21-50,
    # point_head_template.py:49-164) behind a stub head; the point-in-box op they call is answered by the oracle's restatement of
    # the CUDA kernel (pinned against in_hull above), so what is pinned here is the label rule (inside = 1, inside the box enlarged by
    # GT_EXTRA_WIDTH only = ignored) and the loss normalisation
    import oracle as _oracle
    pib = _types.SimpleNamespace(points_in_boxes_gpu=lambda pts, bxs: torch.from_numpy(_oracle.points_in_boxes_gpu(pts.numpy(), bxs.numpy())))
    pns = {"torch": torch, "np": np, "roiaware_pool3d_utils": pib, "common_utils": cu, "box_utils": _types.SimpleNamespace(enlarge_box3d=bns["enlarge_box3d"])}
    stack_t = method("models/dense_heads/point_head_template.py", "PointHeadTemplate", "assign_stack_targets", dict(pns))
    simple_t = method("models/dense_heads/point_head_simple.py", "PointHeadSimple", "assign_targets", dict(pns))
    p_loss = method("models/dense_heads/point_head_template.py", "PointHeadTemplate", "get_cls_layer_loss", dict(pns))
    kp, gb = [], []
    for smp in range(2):
        pts_s, bx = synth.lidar_cloud(51 + smp, 4096, pc_range=np.array(Small.point_cloud_range, np.float32), return_boxes=True)
        kp.append(pts_s[::2, :3])
        gb.append(np.concatenate([bx[:20], np.ones((20, 1), np.float32)], 1).astype(np.float32))
    kp, gb = np.stack(kp).astype(np.float32), np.stack(gb)
    coords = np.concatenate([np.repeat(np.arange(2, dtype=np.float32), kp.shape[1])[:, None], kp.reshape(-1, 3)], 1)
    pstub = _types.SimpleNamespace(num_class=1, model_cfg=_types.SimpleNamespace(TARGET_CONFIG=_types.SimpleNamespace(GT_EXTRA_WIDTH=[0.2, 0.2, 0.2]),
                                                                                  LOSS_CONFIG=_types.SimpleNamespace(LOSS_WEIGHTS={"point_cls_weight": 4.0})),
                                   cls_loss_func=lns["SigmoidFocalClassificationLoss"](alpha=0.25, gamma=2.0))
    pstub.assign_stack_targets = lambda **kw: stack_t(pstub, **kw)
    tdp = simple_t(pstub, {"point_coords": torch.from_numpy(coords), "gt_boxes": torch.from_numpy(gb)})
    logits = rng.normal(-1.0, 2.0, size=(coords.shape[0], 1)).astype(np.float32)
    pstub.forward_ret_dict = {"point_cls_labels": tdp["point_cls_labels"], "point_cls_preds": torch.from_numpy(logits)}
    pl, ptb = p_loss(pstub)
    save("point_head", keypoints=kp, gt=gb, labels=tdp["point_cls_labels"].numpy(), logits=logits, loss=np.float32(pl.item()), positives=np.float32(ptb["point_pos_num"]))

    # ---- BEV backbone -----------------------------------------------------------------------------------------------------------------------
    # BaseBEVBackbone (backbones_2d/base_bev_backbone.py, imported by path) with fv2p.yaml's layout (LAYER_NUMS [5, 5], strides [1, 2],
    # upsample strides [1, 2]) at a sixteenth of the channels: parameters by name + one training-mode forward pass
    class _Cfg(dict):
        __getattr__ = dict.__getitem__
    bb = by_path("ref_bev_backbone", "models/backbones_2d/base_bev_backbone.py")
    torch.manual_seed(11)
    ref_bev = bb.BaseBEVBackbone(_Cfg(LAYER_NUMS=[5, 5], LAYER_STRIDES=[1, 2], NUM_FILTERS=[8, 16], UPSAMPLE_STRIDES=[1, 2], NUM_UPSAMPLE_FILTERS=[16, 16]), 16)
    for prm in ref_bev.parameters():
        if prm.dim() == 1:
            prm.data.uniform_(0.5, 1.5)          # BatchNorm weights / biases away from their 1 / 0 defaults
    x_bev = torch.randn(2, 16, 24, 20)
    y_bev = ref_bev({"spatial_features": x_bev})["spatial_features_2d"]
    save("bev_backbone", x=x_bev.numpy(), y=y_bev.detach().numpy(), **{"p:" + k: v.detach().numpy() for k, v in ref_bev.state_dict().items()})

    # ---- rotated BEV overlap: the reference's OTHER implementation -------------------------------------------------------------------------
    # datasets/kitti/kitti_object_eval_python/rotate_iou.py:17-260 (the evaluation code's quadrilateral intersection: segment crossings +
    # corners inside the other box, vertices sorted around their mean, triangle fan) is numba.cuda device code written in plain Python:
    # with `numba.jit` / `cuda.jit` as identity decorators and `cuda.local.array` as a float32 numpy array it runs as is.  It is
    # independent of iou3d_nms_kernel.cu's algorithm (A16), which the oracle restates, and differs from it only where that kernel is
    # approximate by construction (its point-in-box margin of 1e-2 and 1e-8 guards).  rbbox = (x, y, dx, dy, -heading): the evaluation
    # code turns clockwise, the LiDAR boxes counter-clockwise.
    import sys as _sys
    nb = _types.ModuleType("numba")
    ident = lambda *a, **k: (a[0] if len(a) == 1 and callable(a[0]) and not k else (lambda f: f))
    nb.jit, nb.float32 = ident, np.float32
    nbc = _types.ModuleType("numba.cuda")
    nbc.jit, nbc.local = ident, _types.SimpleNamespace(array=lambda shape, dtype: np.zeros(shape, dtype))
    nb.cuda = nbc
    had = {k: _sys.modules.get(k) for k in ("numba", "numba.cuda")}
    _sys.modules["numba"], _sys.modules["numba.cuda"] = nb, nbc
    try:
        riou = by_path("ref_rotate_iou", "datasets/kitti/kitti_object_eval_python/rotate_iou.py")
    finally:
        for k, v in had.items():
            if v is None:
                _sys.modules.pop(k, None)
            else:
                _sys.modules[k] = v
    rr = np.random.default_rng(77)
    n = 600
    pa, pb = np.zeros((n, 7), np.float32), np.zeros((n, 7), np.float32)
    for bx in (pa, pb):
        bx[:, :2], bx[:, 2] = rr.uniform(-3, 3, (n, 2)), rr.uniform(-1, 1, n)
        bx[:, 3], bx[:, 4], bx[:, 5] = rr.uniform(1.5, 5, n), rr.uniform(1, 2.5, n), rr.uniform(1, 2, n)
        bx[:, 6] = rr.uniform(-np.pi, np.pi, n)
    pb[:40, :2], pb[:40, 6] = pa[:40, :2], pa[:40, 6] + np.float32(np.pi / 2)   # same centre, crossed (identical boxes are degenerate in that code: 0 for 17 of 40 tried)
    pb[40:80, :2] = pa[40:80, :2] + rr.uniform(-0.3, 0.3, (40, 2))   # near-identical centres, different headings
    pb[80:120, 3:5] = pa[80:120, 3:5] * 0.4                       # a small box ...
    pb[80:120, :2] = pa[80:120, :2]                               # ... inside a large one
    pb[120:160, :2] = pa[120:160, :2] + 20                        # far apart
    as_r = lambda q: np.array([q[0], q[1], q[3], q[4], -q[6]], np.float32)
    inter = np.array([riou.inter(as_r(p), as_r(q)) for p, q in zip(pa, pb)], np.float64)
    iou = np.array([riou.devRotateIoUEval(as_r(p), as_r(q), -1) for p, q in zip(pa, pb)], np.float64)
    # 3-D IoU of the same pairs from the evaluation code's d3_box_overlap_kernel (kitti_object_eval_python/eval.py:121-147, compiled out of the
    # file: its module imports numba and the package-relative rotate_iou), fed with the BEV intersections above.  It works on camera boxes
    # (x, y_bottom, z, l, h, w, ry) with y pointing down: a LiDAR box (cx, cy, cz, dx, dy, dz, heading) is (cx, -(cz - dz/2), cy, dx, dz, dy, -heading).
    d3 = extract("datasets/kitti/kitti_object_eval_python/eval.py", ["d3_box_overlap_kernel"], {"numba": nb, "np": np})["d3_box_overlap_kernel"]
    cam = lambda q: np.array([[q[0], -(q[2] - q[5] / 2), q[1], q[3], q[5], q[4], -q[6]]], np.float64)
    iou3d = np.zeros(n, np.float64)
    for i, (p, q) in enumerate(zip(pa, pb)):
        rinc = np.array([[inter[i]]], np.float64)
        d3(cam(p.astype(np.float64)), cam(q.astype(np.float64)), rinc, -1)
        iou3d[i] = rinc[0, 0]
    save("rotate_iou", boxes_a=pa, boxes_b=pb, overlap=inter, iou=iou, iou3d=iou3d)

    # ---- proposal layer of the second stage --------------------------------------------------------------------------------------------------
    # RoIHeadTemplate.proposal_layer (roi_heads/roi_head_template.py:46-99, method compiled out of the class) around
    # model_nms_utils.class_agnostic_nms (model_utils/model_nms_utils.py:6-25, def compiled out of the file): per sample max over classes,
    # top NMS_PRE_MAXSIZE by score, NMS, first NMS_POST_MAXSIZE survivors, zero padding, labels + 1.  `iou3d_nms_utils.nms_gpu` — the op
    # under test elsewhere — is answered by oracle.nms on the already sorted candidates, so the fixture pins the Python composition.
    import oracle as _oracle
    def _nms_gpu(boxes, scores, thresh, pre_maxsize=None, **kwargs):
        keep = _oracle.nms(boxes.numpy(), scores.numpy(), thresh, pre_maxsize=pre_maxsize)
        return torch.from_numpy(keep).long(), None
    pns = {"torch": torch, "iou3d_nms_utils": _types.SimpleNamespace(nms_gpu=_nms_gpu)}
    extract("models/model_utils/model_nms_utils.py", ["class_agnostic_nms"], pns)
    prop_layer = method("models/roi_heads/roi_head_template.py", "RoIHeadTemplate", "proposal_layer", pns)
    class _Nms(dict):
        __getattr__ = dict.__getitem__
    nms_cfg = _Nms(NMS_TYPE="nms_gpu", MULTI_CLASSES_NMS=False, NMS_PRE_MAXSIZE=1000, NMS_POST_MAXSIZE=128, NMS_THRESH=0.8)
    rp = np.random.default_rng(31)
    pb_, ps_ = [], []
    for centres, per in ((60, 25), (300, 5)):          # sample 0: 60 clusters -> fewer than 128 survivors (padding); sample 1: more than 128 (cut)
        base = random_boxes(rp, centres, spread=30.0)
        bx = np.repeat(base, per, 0) + rp.normal(0, 0.05, (centres * per, 7)).astype(np.float32)
        pb_.append(bx[rp.permutation(len(bx))].astype(np.float32))
        ps_.append(rp.normal(0, 2, len(bx)).astype(np.float32))
    pbx, psc = np.stack(pb_), np.stack(ps_)
    bd = {"batch_size": 2, "batch_box_preds": torch.from_numpy(pbx), "batch_cls_preds": torch.from_numpy(psc)[..., None]}
    bd = prop_layer(_types.SimpleNamespace(), bd, nms_cfg)
    save("proposal_layer", boxes=pbx, scores=psc, rois=bd["rois"].numpy(), roi_scores=bd["roi_scores"].numpy(), roi_labels=bd["roi_labels"].numpy(),
         nms_pre=1000, nms_post=128, nms_thresh=np.float32(0.8))

    # ---- voxel-to-point decoder ------------------------------------------------------------------------------------------------------------------
    # ResidualVoxelToPointDecoder + LateralBottomResBlock (backbones_3d/pfe/residual_v2p_decoder.py:46-313, classes compiled out of the file:
    # its module imports the CUDA extensions) around pointnet2_batch_utils.top3_interpolate (pointnet2_batch/pointnet2_utils.py:292-326, def
    # compiled out of the file).  The three ops underneath — furthest_point_sample, three_nn, three_interpolate — are answered by the
    # oracle, so the fixture pins the Python of the paper's decoder: key points per sample incl. the short-cloud rule (:218-220), voxel
    # centres per level, inverse-distance weights, the residual blocks and their parameter names.  fv2p.yaml's block layout at an eighth
    # of the channels, 1 024 key points, one cloud shorter than that.
    def _three_nn(unknown, known):
        d2, idx = _oracle.three_nn_batch(unknown.numpy(), known.numpy())
        return torch.sqrt(torch.from_numpy(d2)), torch.from_numpy(idx)          # pointnet2_utils.py:92-93
    def _three_interpolate(features, idx, weight):
        return torch.from_numpy(_oracle.three_interpolate_batch(features.detach().numpy(), idx.numpy(), weight.detach().numpy()))
    def _fps(xyz, npoint):
        return torch.from_numpy(_oracle.furthest_point_sample(xyz.numpy(), npoint)[0])
    tns = {"torch": torch, "three_nn": _three_nn, "three_interpolate": _three_interpolate}
    extract("ops/pointnet2/pointnet2_batch/pointnet2_utils.py", ["top3_interpolate"], tns)
    dns = {"torch": torch, "nn": torch.nn, "common_utils": cu, "pointnet2_batch_utils": _types.SimpleNamespace(top3_interpolate=tns["top3_interpolate"]),
           "pointnet2_stack_utils": _types.SimpleNamespace(furthest_point_sample=_fps)}
    extract("models/backbones_3d/pfe/residual_v2p_decoder.py", ["LateralBottomResBlock", "ResidualVoxelToPointDecoder"], dns)
    class _C(dict):
        __getattr__ = dict.__getitem__
    blk = lambda stride, lat, out: _C(LATERAL_DOWNSAMPLE_FACTOR=stride, BOTTOM_DOWNSAMPLE_FACTOR=1, LATERAL_CHANNELS=lat, OUT_CHANNELS=out, NSAMPLE=3)
    dec_levels = (("x_conv4", 8, 16, 32), ("x_conv3", 4, 8, 24), ("x_conv2", 2, 4, 20), ("x_conv1", 1, 2, 16))
    dcfg = _C(POINT_SOURCE="raw_points", SAMPLE_METHOD="FPS", NUM_KEYPOINTS=1024, FEATURES_SOURCE=["x_conv4", "x_conv3", "x_conv2", "x_conv1"],
              INIT_BLOCK=_C(blk(8, 16, 16), SOURCE="x_conv4"), DECODE_BLOCKS=_C({n: blk(st, lat, out) for n, st, lat, out in dec_levels}),
              OUT_BLOCK=_C(OUT_CHANNELS=16, NSAMPLE=3))
    d_vs, d_rng = [0.05, 0.05, 0.1], [0.0, -12.8, -3.0, 25.6, 12.8, 1.0]           # grid 512 x 512 x 40
    torch.manual_seed(23)
    ref_dec = dns["ResidualVoxelToPointDecoder"](dcfg, d_vs, d_rng)
    for prm in ref_dec.parameters():
        if prm.dim() == 1:
            prm.data.uniform_(0.5, 1.5)
    rd = np.random.default_rng(41)
    pts, lvl = [], {}
    for b_i, n_pts in enumerate((3000, 900)):                                      # the second cloud is shorter than NUM_KEYPOINTS
        xyz = np.stack([rd.uniform(0.5, 25.0, n_pts), rd.uniform(-12.0, 12.0, n_pts), rd.uniform(-2.5, 0.5, n_pts)], 1)
        pts.append(np.concatenate([np.full((n_pts, 1), b_i), xyz, rd.random((n_pts, 1))], 1).astype(np.float32))
    pts_all = np.concatenate(pts)
    for (name, stride, lat, _), n_vox in zip(dec_levels, (400, 900, 1500, 2500)):
        gz, gy, gx = (41 + stride - 1) // stride, 512 // stride, 512 // stride
        rows = []
        for b_i in range(2):
            cells = rd.choice(gz * gy * gx, n_vox, replace=False)
            rows.append(np.stack([np.full(n_vox, b_i), cells // (gy * gx), (cells // gx) % gy, cells % gx], 1))
        lvl[name] = (np.concatenate(rows).astype(np.int32), rd.standard_normal((2 * n_vox, lat)).astype(np.float32))
    bdict = {"batch_size": 2, "points": torch.from_numpy(pts_all),
             "multi_scale_3d_features": {k: _types.SimpleNamespace(indices=torch.from_numpy(i), features=torch.from_numpy(f)) for k, (i, f) in lvl.items()}}
    ref_dec.train()
    key_ref = ref_dec.get_sampled_points(bdict)
    od = ref_dec(bdict)
    save("v2p_decoder", points=pts_all, keypoints=key_ref.numpy(), point_coords=od["point_coords"].numpy(), point_features=od["point_features"].detach().numpy(),
         voxel_size=np.array(d_vs, np.float32), point_cloud_range=np.array(d_rng, np.float32),
         **{f"ind:{k}": i for k, (i, f) in lvl.items()}, **{f"feat:{k}": f for k, (i, f) in lvl.items()},
         **{"p:" + k: v.detach().numpy() for k, v in ref_dec.state_dict().items()})

    # ---- grid set abstraction of the second stage -------------------------------------------------------------------------------------------------
    # PointnetSAModuleMSG + QueryAndGroup (pointnet2_batch/pointnet2_modules.py:10-98, pointnet2_utils.py:231-264; classes compiled out of their
    # files) with ball_query / grouping_operation answered by the oracle: the module IoUGuidedRoIHead builds for its grid points
    # (iouguided_roi_head.py:52-76: radii 0.8 / 1.6, 16 / 32 samples, mlps [C -> 64 -> 64] x 2, use_xyz, no BatchNorm; C = 32 instead of 128 input channels) on 4 RoIs of 512
    # canonical points and the 216 grid points of each, one RoI empty (all points at the origin, :188).  Pins what the fused kernel
    # (first layer per point / per centre, gather + second layer + max in one launch) has to reproduce.
    from typing import List as _List, Tuple as _Tuple
    def _ball_query(radius, nsample, xyz, new_xyz):
        return torch.from_numpy(_oracle.ball_query_batch(radius, nsample, xyz.numpy(), new_xyz.numpy()))
    def _grouping(features, idx):
        return torch.from_numpy(_oracle.group_points_batch(features.detach().numpy(), idx.numpy()))
    uns = {"torch": torch, "nn": torch.nn, "Tuple": _Tuple, "ball_query": _ball_query, "grouping_operation": _grouping}
    extract("ops/pointnet2/pointnet2_batch/pointnet2_utils.py", ["QueryAndGroup"], uns)
    mns = {"torch": torch, "nn": torch.nn, "F": torch.nn.functional, "List": _List, "pointnet2_utils": _types.SimpleNamespace(QueryAndGroup=uns["QueryAndGroup"])}
    extract("ops/pointnet2/pointnet2_batch/pointnet2_modules.py", ["_PointnetSAModuleBase", "PointnetSAModuleMSG"], mns)
    torch.manual_seed(29)
    ref_sa = mns["PointnetSAModuleMSG"](npoint=216, radii=[0.8, 1.6], nsamples=[16, 32], mlps=[[32, 64, 64], [32, 64, 64]], use_xyz=True, bn=False)
    rs = np.random.default_rng(53)
    n_roi = 4
    sa_xyz = (rs.uniform(-1, 1, (n_roi, 512, 3)) * np.array([2.6, 1.4, 1.0])).astype(np.float32)
    sa_xyz[2, 300:] = sa_xyz[2, :212]                                   # a RoI pooled from fewer points than slots: repeats
    sa_feat = rs.standard_normal((n_roi, 32, 512)).astype(np.float32)
    sa_feat[2, :, 300:] = sa_feat[2, :, :212]
    sa_xyz[3], sa_feat[3] = 0, 0                                        # an empty RoI
    sizes = np.array([[3.9, 1.6, 1.56]], np.float32) * rs.uniform(0.8, 1.2, (n_roi, 3)).astype(np.float32)
    gi = (np.stack(np.meshgrid(np.arange(6), np.arange(6), np.arange(6), indexing="ij"), -1).reshape(1, 216, 3).astype(np.float32) + 0.5) / 6
    sa_grid = (gi * sizes[:, None] - sizes[:, None] / 2).astype(np.float32)
    _, sa_out = ref_sa(torch.from_numpy(sa_xyz), torch.from_numpy(sa_feat), torch.from_numpy(sa_grid))
    save("sa_grid", xyz=sa_xyz, features=sa_feat, grid=sa_grid, out=sa_out.detach().numpy(), **{"p:" + k: v.detach().numpy() for k, v in ref_sa.state_dict().items()})

    # ---- point stream of the second stage: RoI point pooling in the RoI frame ------------------------------------------------------------------------
    # IoUGuidedRoIHead.roipool3d_gpu (roi_heads/iouguided_roi_head.py:144-193, method compiled out of the class) around RoIPointPool3dFunction.forward
    # (roipoint_pool3d/roipoint_pool3d_utils.py:31-63, class compiled out of the file) with `roipoint_pool3d_cuda.forward` answered by the oracle:
    # per-point [score, depth / DEPTH_NORMALIZER - 0.5, features], boxes enlarged by POOL_EXTRA_WIDTH (a list: enlarge_box3d), pooled points
    # moved into the RoI frame (centre, then rotation by -heading), empty RoIs zeroed.
    def _pool_fwd(points, boxes, feats, pooled, flag):
        pf, fl = _oracle.roipoint_pool3d(points.numpy(), feats.numpy(), boxes.numpy(), pooled.shape[2])
        pooled.copy_(torch.from_numpy(pf)), flag.copy_(torch.from_numpy(fl))
    rpn = {"torch": torch, "nn": torch.nn, "Function": torch.autograd.Function, "box_utils": _types.SimpleNamespace(enlarge_box3d=bns["enlarge_box3d"]),
           "roipoint_pool3d_cuda": _types.SimpleNamespace(forward=_pool_fwd)}
    extract("ops/roipoint_pool3d/roipoint_pool3d_utils.py", ["RoIPointPool3d", "RoIPointPool3dFunction"], rpn)
    hns = {"torch": torch, "common_utils": cu}
    pool_m = method("models/roi_heads/iouguided_roi_head.py", "IoUGuidedRoIHead", "roipool3d_gpu", hns)
    rq = np.random.default_rng(61)
    n_key, n_roi, c_pt = 2048, 16, 8
    key_xyz = np.stack([rq.uniform(0, 40, (2, n_key)), rq.uniform(-20, 20, (2, n_key)), rq.uniform(-2.5, 0.5, (2, n_key))], -1).astype(np.float32)
    rois_p = np.stack([random_boxes(rq, n_roi, spread=18.0) for _ in range(2)])
    rois_p[..., 0] += 20
    rois_p[:, -1, :3] = (200.0, 200.0, 0.0)                                     # nothing near: empty flag
    rois_p[:, 0, 3:6] *= 0.4                                                    # a small box: fewer points than slots, repeated to fill them
    pstub2 = _types.SimpleNamespace(model_cfg=_C(ROI_POINT_POOL=_C(DEPTH_NORMALIZER=70.0)),
                                    roipoint_pool3d_layer=rpn["RoIPointPool3d"](num_sampled_points=64, pool_extra_width=[0.4, 0.4, 0.4]))
    bd2 = {"batch_size": 2, "point_coords": torch.from_numpy(np.concatenate([np.repeat(np.arange(2, dtype=np.float32), n_key)[:, None], key_xyz.reshape(-1, 3)], 1)),
           "point_features": torch.from_numpy(rq.standard_normal((2 * n_key, c_pt)).astype(np.float32)),
           "point_cls_scores": torch.from_numpy(rq.random(2 * n_key).astype(np.float32))}
    pooled_ref = pool_m(pstub2, bd2, torch.from_numpy(rois_p))
    save("roi_point_pool", keypoints=key_xyz, point_features=bd2["point_features"].numpy(), point_scores=bd2["point_cls_scores"].numpy(), rois=rois_p,
         pooled=pooled_ref.numpy())

    # ---- second stage, forward_single_loop as a whole -----------------------------------------------------------------------------------------------------
    # The reference's own IoUGuidedRoIHead class (roi_heads/iouguided_roi_head.py:11-304: __init__, roipool3d_gpu, the grid points, forward_single_loop;
    # class compiled out of its file) over the reference classes pinned one by one above (RoIPointPool3d, PointnetSAModuleMSG), its
    # CornerGeometryEncodeModule / FeatureFusionModule (feature_adaptor/nn_modules.py, imported by path), BEVGridPooling + bilinear_interpolate_torch
    # (backbones_3d/pfe/bev_grid_pooling.py, compiled out of the file) and boxes_to_CTcorners_3d (utils/box_utils.py:56-87).  The base class is
    # reduced to what __init__ reads (make_fc_layers of roi_withiou_head_template.py:29-43, num_class, the coder's code size); the CUDA ops are the
    # oracle's.  fv2p.yaml's head at reduced widths: 3^3 grid points, 128 pooled points, no dropout (DP_RATIO 0), training-mode BatchNorm.
    nnm = by_path("ref_roi_nn_modules", "models/roi_heads/feature_adaptor/nn_modules.py")
    gns = extract("models/backbones_3d/pfe/bev_grid_pooling.py", ["bilinear_interpolate_torch", "BEVGridPooling"], {"torch": torch, "nn": torch.nn})
    ctn = extract("utils/box_utils.py", ["boxes_to_CTcorners_3d"], {"torch": torch, "np": np, "common_utils": cu})
    class _HeadBase(torch.nn.Module):
        def __init__(self, num_class, model_cfg):
            super().__init__()
            self.num_class, self.model_cfg, self.box_coder = num_class, model_cfg, _types.SimpleNamespace(code_size=7)
    _HeadBase.make_fc_layers = method("models/roi_heads/roi_withiou_head_template.py", "RoIWithIoUHeadTemplate", "make_fc_layers", {"nn": torch.nn})
    ins = {"torch": torch, "nn": torch.nn, "pointnet2_modules": _types.SimpleNamespace(PointnetSAModuleMSG=mns["PointnetSAModuleMSG"]),
           "roipoint_pool3d_utils": _types.SimpleNamespace(RoIPointPool3d=rpn["RoIPointPool3d"]), "common_utils": cu,
           "box_utils": _types.SimpleNamespace(boxes_to_CTcorners_3d=ctn["boxes_to_CTcorners_3d"]), "RoIWithIoUHeadTemplate": _HeadBase,
           "CornerGeometryEncodeModule": nnm.CornerGeometryEncodeModule, "FeatureFusionModule": nnm.FeatureFusionModule, "BEVGridPooling": gns["BEVGridPooling"]}
    extract("models/roi_heads/iouguided_roi_head.py", ["IoUGuidedRoIHead"], ins)
    hcfg = _C(BEV_GRID_POOL=_C(IN_CHANNELS=32, OUT_CHANNELS=16), ROI_POINT_POOL=_C(NUM_SAMPLED_POINTS=128, POOL_EXTRA_WIDTH=[1.0, 1.0, 1.0], DEPTH_NORMALIZER=70.0),
              USE_BN=False, XYZ_UP_LAYER=[32, 32], ROI_GRID_POOL=_C(GRID_SIZE=3, SA_CONFIG=_C(NPOINTS=[27], RADIUS=[[0.8, 1.6]], NSAMPLE=[[16, 32]], MLPS=[[[64, 64], [64, 64]]])),
              GRID_INTERACT=_C(INTERACT_FILTERS=[32, 32]), DP_RATIO=0, CGE_MODULE=_C(UP_FILTERS=[16, 16], INTERACT_FILTERS=[32]), FUSE_FILTERS=[32],
              CLS_FC=[32, 32], REG_FC=[32, 32], TARGET_CONFIG=_C(CLS_SCORE_TYPE="roi_iou"))
    h_vs, h_rng = [0.05, 0.05, 0.1], [0.0, -6.4, -3.0, 12.8, 6.4, 1.0]
    torch.manual_seed(37)
    ref_head = ins["IoUGuidedRoIHead"](input_channels=32, model_cfg=hcfg, point_cloud_range=h_rng, voxel_size=h_vs, num_class=1)
    with torch.no_grad():
        ref_head.reg_layers[-1].weight.normal_(0, 0.1)            # away from its 0.001 init: the residuals should carry signal
        for prm in ref_head.parameters():
            if prm.dim() == 1:
                prm.uniform_(0.5, 1.5)
    ref_head.train()
    rh = np.random.default_rng(67)
    n_key2, n_roi2 = 1024, 8
    key2 = np.stack([rh.uniform(0.2, 12.6, (2, n_key2)), rh.uniform(-6.2, 6.2, (2, n_key2)), rh.uniform(-2.5, 0.5, (2, n_key2))], -1).astype(np.float32)
    rois2 = np.stack([random_boxes(rh, n_roi2, spread=4.5) for _ in range(2)])
    rois2[..., 0] += 6.4
    rois2[0, -1, :2] = (60.0, 60.0)                                # off the map and away from every point
    bev2 = rh.standard_normal((2, 32, 32, 32)).astype(np.float32)
    bd3 = {"batch_size": 2, "point_coords": torch.from_numpy(np.concatenate([np.repeat(np.arange(2, dtype=np.float32), n_key2)[:, None], key2.reshape(-1, 3)], 1)),
           "point_features": torch.from_numpy(rh.standard_normal((2 * n_key2, 32)).astype(np.float32)),
           "point_cls_scores": torch.from_numpy(rh.random(2 * n_key2).astype(np.float32)),
           "spatial_features_before_head": torch.from_numpy(bev2), "spatial_features_stride": 8}
    r_cls, r_reg, r_iou = ref_head.forward_single_loop(bd3, torch.from_numpy(rois2))
    save("roi_head_forward", keypoints=key2, point_features=bd3["point_features"].numpy(), point_scores=bd3["point_cls_scores"].numpy(), rois=rois2, bev=bev2,
         voxel_size=np.array(h_vs, np.float32), point_cloud_range=np.array(h_rng, np.float32),
         rcnn_cls=r_cls.detach().numpy(), rcnn_reg=r_reg.detach().numpy(), rcnn_iou=r_iou.detach().numpy(),
         **{"p:" + k: v.detach().numpy() for k, v in ref_head.state_dict().items()})

    # ---- the reference's own sparse test data ---------------------------------------------------------------------------------------------
    # spconv/test_utils.py:144-193 generate_sparse_data (imported by path: numpy only; upstream spconv's test_conv.py — SparseConv3d against
    # nn.Conv3d on this generator's `features_dense` — was not vendored with it, tests/ restate that test): unique random cells per sample,
    # uniform features, and the dense tensor the generator scatters itself.  Indices come out as (z, y, x, batch).
    tu = by_path("ref_spconv_test_utils", "ops/spconv/test_utils.py")
    np.random.seed(484)
    sd = tu.generate_sparse_data([19, 18, 17], [1500, 1300], 16)
    save("sparse_data", features=sd["features"], indices=sd["indices"], features_dense=sd["features_dense"], spatial_shape=np.array([19, 18, 17]))


if __name__ == "__main__":
    main()
