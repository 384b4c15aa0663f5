"""Build-container-only: the reference's model package imports unchanged against this repo's `pcdet.ops`.

A temporary tree is assembled from symlinks — every entry of /root/reference/pcdet except `ops`, which points at this
repo's pcdet/ops — and a fresh interpreter imports the reference's backbones, decoder, heads and detectors from it
(the import surface pcdet/models/** expects); the reference's two sparse backbones, built on this repo's spconv
modules, have the parameter names and shapes of the replay harness's re-declarations and, with one state dict, compute the same features at every level (both run on the host through the oracle's conv shim).  Third-party packages the image lacks (cv2, numba, mmcv,
shapely, easydict, ...) are replaced by empty stand-in modules inside that interpreter: they are not part of the boundary.
Then the reference's own ResidualVoxelToPointDecoder and IoUGuidedRoIHead (unmodified classes) run on the host as consumers of this package's
Python layer, every C-ABI call answered by the oracle, and reproduce the fixtures written from the same classes with the ops stubbed out
directly.  Skipped where /root/reference does not exist (the GPU box)."""
import os
import subprocess
import sys

import pytest

REF = "/root/reference/pcdet"
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PRELUDE = r'''
import importlib, sys, types
for name in ["cv2", "numba", "easydict", "shapely", "shapely.geometry", "mmcv", "mmcv.cnn", "SharedArray", "tensorboardX", "kornia",
             "skimage", "skimage.io", "torchsparse", "torchsparse.nn"]:
    try:
        importlib.import_module(name)
    except Exception:
        m = types.ModuleType(name)
        m.__path__ = []
        sys.modules[name] = m
nb = sys.modules["numba"]
if not hasattr(nb, "jit"):
    nb.jit = lambda *a, **k: (lambda f: f)
ed = sys.modules["easydict"]
if not hasattr(ed, "EasyDict"):
    def _attr(self, key):
        try:
            return self[key]
        except KeyError:
            raise AttributeError(key)
    ed.EasyDict = type("EasyDict", (dict,), {"__getattr__": _attr, "__setattr__": dict.__setitem__})
sg = sys.modules["shapely.geometry"]
if not hasattr(sg, "Polygon"):
    sg.Polygon = object
mc = sys.modules["mmcv.cnn"]
if not hasattr(mc, "CONV_LAYERS"):
    mc.CONV_LAYERS = type("R", (), {"register_module": lambda self, *a, **k: (lambda c: c)})()
'''

SCRIPT = PRELUDE + r'''import pcdet.ops.spconv as spconv
assert "from-voxel-to-point_amd" in os.path.realpath(spconv.__file__) if (os := __import__("os")) else True
mods = ["pcdet.models.backbones_3d.spconv_backbone", "pcdet.models.backbones_3d.pfe.residual_v2p_decoder",
        "pcdet.models.backbones_3d.pfe.bev_grid_pooling", "pcdet.models.backbones_2d.dcn_bev_backbone",
        "pcdet.models.dense_heads.point_head_simple", "pcdet.models.dense_heads.center_af_head_single",
        "pcdet.models.roi_heads.iouguided_roi_head", "pcdet.models.detectors.fv2p", "pcdet.models.detectors.mgaf_3dssd",
        "pcdet.datasets.processor.data_processor"]
for m in mods:
    importlib.import_module(m)
from pcdet.models.backbones_3d.spconv_backbone import VoxelResBackBone8x, VoxelBackBone8x
from pcdet.models.backbones_3d.pfe.residual_v2p_decoder import ResidualVoxelToPointDecoder
# the reference's own backbone class builds on this repo's spconv modules (layer list, indice keys, parameters)
net = VoxelResBackBone8x(ed.EasyDict(), input_channels=4, grid_size=__import__("numpy").array([1408, 1600, 40]))
n_conv = sum(isinstance(m, spconv.SparseConvolution) for m in net.modules())
assert n_conv == 21, n_conv
# ... and the replay harness re-declares the same networks: parameter / buffer names and shapes agree one to one
import numpy as np
from fv2p_harness import backbone as hb
for ref_cls, own_cls in ((VoxelResBackBone8x, hb.VoxelResBackBone8x), (VoxelBackBone8x, hb.VoxelBackBone8x)):
    ref_sd = {k: tuple(v.shape) for k, v in ref_cls(ed.EasyDict(), input_channels=4, grid_size=np.array([1408, 1600, 40])).state_dict().items()}
    own_sd = {k: tuple(v.shape) for k, v in own_cls(4, [1408, 1600, 40]).state_dict().items()}
    assert ref_sd == own_sd, (sorted(set(ref_sd.items()) ^ set(own_sd.items()))[:8])
# ... and compute the same function: the reference class and the harness class with one state dict, both run on the host through
# the oracle's conv shim (oracle.spconv_cpu.cpu_mirror swaps every SparseConvolution of a module tree), training-mode BatchNorm
import torch
from oracle.spconv_cpu import cpu_mirror
rng = np.random.default_rng(3)
grid = [96, 80, 24]                                              # x, y, z cells; sparse_shape = [25, 80, 96]
cells = rng.choice(2 * 24 * 80 * 96, 1500, replace=False)
coords = np.stack([cells // (24 * 80 * 96), (cells // (80 * 96)) % 24, (cells // 96) % 80, cells % 96], 1).astype(np.int32)
coords = coords[np.lexsort((coords[:, 3], coords[:, 2], coords[:, 1], coords[:, 0]))]
feats = torch.from_numpy(rng.standard_normal((1500, 4)).astype(np.float32))
for ref_cls, own_cls in ((VoxelResBackBone8x, hb.VoxelResBackBone8x), (VoxelBackBone8x, hb.VoxelBackBone8x)):
    torch.manual_seed(5)
    ref_net = ref_cls(ed.EasyDict(), input_channels=4, grid_size=np.array(grid))
    own_net = own_cls(4, grid)
    own_net.load_state_dict(ref_net.state_dict(), strict=True)
    ref_cpu, own_cpu = cpu_mirror(ref_net).train(), cpu_mirror(own_net).train()
    bd = ref_cpu({"voxel_features": feats.clone(), "voxel_coords": torch.from_numpy(coords), "batch_size": 2})
    out, levels = own_cpu(feats.clone(), torch.from_numpy(coords), 2)
    enc = bd["encoded_spconv_tensor"]
    assert torch.equal(enc.indices, out.indices) and list(enc.spatial_shape) == list(out.spatial_shape)
    assert float((enc.features - out.features).abs().max()) < 1e-5 * float(enc.features.abs().max()), ref_cls.__name__
    for name, lvl in bd["multi_scale_3d_features"].items():
        assert torch.equal(lvl.indices, levels[name].indices), name
        assert float((lvl.features - levels[name].features).abs().max()) < 1e-5 * float(lvl.features.abs().max()), name
# ---- the reference's own decoder and second-stage head, unmodified, as consumers of this package's Python layer -------------------------
# Every C-ABI call is answered by the oracle (oracle.backend), so the reference classes run on the host through pcdet.ops' wrappers
# (argument order, layouts, dtypes, autograd Functions) exactly as they would on the HIP library; the expected outputs are the fixtures
# oracle/gen_golden_pyref.py wrote from the same classes with the ops stubbed out directly (tests/golden/pyref_v2p_decoder.npz,
# pyref_roi_head_forward.npz) — the two routes differ only by this package's glue.
import types, yaml
from oracle.backend import oracle_backend
torch.Tensor.cuda = lambda self, *a, **k: self                    # loss_utils.WeightedSmoothL1Loss moves its code weights in __init__
def ED(d):
    return ed.EasyDict({k: ED(v) for k, v in d.items()}) if isinstance(d, dict) else d
gold = lambda name: np.load(os.path.join(os.environ["FV2P_GOLDEN"], "pyref_%s.npz" % name))
t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
yml = yaml.safe_load(open("/root/reference/tools/cfgs/kitti_models/FV2P/fv2p.yaml"))["MODEL"]

g = gold("v2p_decoder")
dcfg = ED(yml["POST_PFE"])
dcfg.NUM_KEYPOINTS = 1024
for blk, (lat, out_c) in {"x_conv4": (16, 32), "x_conv3": (8, 24), "x_conv2": (4, 20), "x_conv1": (2, 16)}.items():
    dcfg.DECODE_BLOCKS[blk].LATERAL_CHANNELS, dcfg.DECODE_BLOCKS[blk].OUT_CHANNELS = lat, out_c
dcfg.INIT_BLOCK.LATERAL_CHANNELS = dcfg.INIT_BLOCK.OUT_CHANNELS = dcfg.OUT_BLOCK.OUT_CHANNELS = 16
dec = ResidualVoxelToPointDecoder(dcfg, [float(v) for v in g["voxel_size"]], [float(v) for v in g["point_cloud_range"]]).train()
dec.load_state_dict({k[2:]: t(g[k]) for k in g.files if k.startswith("p:")}, strict=True)
bd = {"batch_size": 2, "points": t(g["points"]),
      "multi_scale_3d_features": {n: types.SimpleNamespace(indices=t(g["ind:" + n]), features=t(g["feat:" + n])) for n in ("x_conv1", "x_conv2", "x_conv3", "x_conv4")}}
with oracle_backend():
    bd = dec(bd)
assert torch.equal(bd["point_coords"], t(g["point_coords"])), "key points through this package's furthest_point_sample differ"
assert float((bd["point_features"] - t(g["point_features"])).abs().max()) < 1e-5 * float(np.abs(g["point_features"]).max())

from pcdet.models.roi_heads.iouguided_roi_head import IoUGuidedRoIHead
g = gold("roi_head_forward")
hcfg = ED(yml["ROI_HEAD"])
hcfg.BEV_GRID_POOL.IN_CHANNELS, hcfg.BEV_GRID_POOL.OUT_CHANNELS = 32, 16
hcfg.ROI_POINT_POOL.NUM_SAMPLED_POINTS, hcfg.ROI_POINT_POOL.POOL_EXTRA_WIDTH = 128, [1.0, 1.0, 1.0]
hcfg.XYZ_UP_LAYER, hcfg.CLS_FC, hcfg.REG_FC, hcfg.FUSE_FILTERS, hcfg.DP_RATIO = [32, 32], [32, 32], [32, 32], [32], 0
hcfg.ROI_GRID_POOL.GRID_SIZE, hcfg.ROI_GRID_POOL.SA_CONFIG.NPOINTS = 3, [27]
hcfg.GRID_INTERACT.INTERACT_FILTERS, hcfg.CGE_MODULE.UP_FILTERS, hcfg.CGE_MODULE.INTERACT_FILTERS = [32, 32], [16, 16], [32]
head = IoUGuidedRoIHead(input_channels=32, model_cfg=hcfg, point_cloud_range=[float(v) for v in g["point_cloud_range"]],
                        voxel_size=[float(v) for v in g["voxel_size"]], num_class=1).train()
missing = head.load_state_dict({k[2:]: t(g[k]) for k in g.files if k.startswith("p:")}, strict=False)
assert not missing.unexpected_keys and all(k.startswith("reg_loss_func") for k in missing.missing_keys), missing
n_key = g["keypoints"].shape[1]
bd = {"batch_size": 2, "point_coords": torch.cat((torch.arange(2.).repeat_interleave(n_key)[:, None], t(g["keypoints"]).view(-1, 3)), 1),
      "point_features": t(g["point_features"]), "point_cls_scores": t(g["point_scores"]),
      "spatial_features_before_head": t(g["bev"]), "spatial_features_stride": 8}
with oracle_backend():
    r_cls, r_reg, r_iou = head.forward_single_loop(bd, t(g["rois"]))
for got, want in ((r_cls, g["rcnn_cls"]), (r_reg, g["rcnn_reg"]), (r_iou, g["rcnn_iou"])):
    assert float((got.detach() - t(want)).abs().max()) < 1e-5 * max(1.0, float(np.abs(want).max()))
# the reference's PointHeadSimple: targets through this package's roiaware points_in_boxes_gpu wrapper (point_head_template.py:84-91)
from pcdet.models.dense_heads.point_head_simple import PointHeadSimple
g = gold("point_head")
ph = PointHeadSimple(num_class=1, input_channels=16, model_cfg=ED(yml["POINT_HEAD"]))
kp = g["keypoints"]
coords = torch.cat((torch.arange(float(kp.shape[0])).repeat_interleave(kp.shape[1])[:, None], t(kp).view(-1, 3)), 1)
with oracle_backend():
    td = ph.assign_targets({"point_coords": coords, "gt_boxes": t(g["gt"]), "batch_size": kp.shape[0]})
assert np.array_equal(td["point_cls_labels"].numpy(), g["labels"])
# the Waymo-shaped workload (bench.py --workload fv2p-waymo, BASELINE configs[4]): the reference's detector built from its
# waymo_fv2p_e30.yaml has the parameters and buffers of the harness's FV2PWaymoConfig detector, name for name and shape for shape
from pcdet.models.detectors.fv2p import FromVoxelToPoint
from fv2p_harness import fv2p_model as fm
wy = yaml.safe_load(open("/root/reference/tools/cfgs/waymo_models/FV2P/waymo_fv2p_e30.yaml"))
W = fm.FV2PWaymoConfig
dataset = types.SimpleNamespace(class_names=wy["CLASS_NAMES"], point_feature_encoder=types.SimpleNamespace(num_point_features=W.num_point_features),
                                grid_size=np.array(W.grid_size), point_cloud_range=np.array(W.point_cloud_range, np.float32), voxel_size=list(W.voxel_size))
ref_w = FromVoxelToPoint(model_cfg=ED(wy["MODEL"]), num_class=1, dataset=dataset)
ren = lambda k: (k.replace("roi_head.CGE_module.corners_up_layer.", "roi_head.cge_up.").replace("roi_head.CGE_module.corners_inter_layer.", "roi_head.cge_inter.")
                  .replace("roi_head.feature_fusion.fuse_layer.", "roi_head.feature_fusion."))
ref_sd = {ren(k): tuple(v.shape) for k, v in ref_w.state_dict().items() if k != "global_step"}
own_sd = {k: tuple(v.shape) for k, v in fm.FV2PDetector(W).state_dict().items()}
assert ref_sd == own_sd and len(own_sd) == 380, sorted(set(ref_sd.items()) ^ set(own_sd.items()))[:8]
assert tuple(ref_w.dense_head.anchors[0].shape) == (1, 188, 188, 1, 2, 7)           # Vehicle only: 2 anchors per BEV cell
# the MGAF-3DSSD layer replay (bench.py --workload mgaf, BASELINE configs[3]): the reference's MGAF3DSSD detector built from
# mgaf-3dssd_3classes.yaml (VoxelResBackBone8x, DCNBEVBackbone on this repo's MdeformConvBlock, CenterAFHeadSingle) against the
# harness's MGAFDetector: 361 parameters / buffers, same names (the harness keeps the seven heads in a ModuleDict) and shapes
from pcdet.models.detectors.mgaf_3dssd import MGAF3DSSD
from fv2p_harness import mgaf_model as mm
my = yaml.safe_load(open("/root/reference/tools/cfgs/kitti_models/MGAF-3DSSD/mgaf-3dssd_3classes.yaml"))
dataset = types.SimpleNamespace(class_names=my["CLASS_NAMES"], point_feature_encoder=types.SimpleNamespace(num_point_features=4), grid_size=np.array([1408, 1600, 40]),
                                point_cloud_range=np.array([0, -40, -3, 70.4, 40, 1], np.float32), voxel_size=[0.05, 0.05, 0.1])
ref_m = MGAF3DSSD(model_cfg=ED(my["MODEL"]), num_class=len(my["CLASS_NAMES"]), dataset=dataset)
ref_sd = {k: tuple(v.shape) for k, v in ref_m.state_dict().items() if k != "global_step"}
own_sd = {k.replace("dense_head.heads.", "dense_head."): tuple(v.shape) for k, v in mm.MGAFDetector(mm.MGAFConfig).state_dict().items()}
assert ref_sd == own_sd and len(own_sd) == 361, sorted(set(ref_sd.items()) ^ set(own_sd.items()))[:8]
# ... and its dense part computes the same seven head maps: the reference's DCNBEVBackbone + CenterAFHeadSingle (on this package's
# MdeformConvBlock / ModulatedDeformConv, DCN forward answered by the oracle) against the harness's re-declarations, one state dict,
# offset / mask predictors moved off their zero initialisation, training-mode BatchNorm
torch.manual_seed(3)
with torch.no_grad():
    for sub in ref_m.modules():
        if hasattr(sub, "conv_offset_mask"):
            sub.conv_offset_mask.weight.normal_(0, 0.05)
head_names = dict(mm.MGAFConfig.heads)
own_m = mm.MGAFDetector(mm.MGAFConfig, offset_init_std=0.0)
own_m.load_state_dict({(k.replace("dense_head.", "dense_head.heads.") if k.split(".")[1] in head_names else k): v
                       for k, v in ref_m.state_dict().items() if k != "global_step"}, strict=True)
ref_m.train(), own_m.train()
ref_m.dense_head.assign_targets = lambda gt_boxes: {}
x = torch.randn(2, 256, 24, 20)
with oracle_backend(), torch.no_grad():
    dd = ref_m.backbone_2d({"spatial_features": x.clone()})
    dd["gt_boxes"] = None
    try:
        ref_m.dense_head(dd)
    except KeyError as e:
        assert e.args[0] == "ind_target"       # the target-based box generation that follows the predictions in training mode
    preds = own_m.dense_head(own_m.backbone_2d(x.clone()))
for name, got in preds.items():
    want = ref_m.dense_head.forward_ret_dict[name + "_pred"]
    assert got.shape == want.shape and float((got - want).abs().max()) <= 1e-5 * float(want.abs().max()), name
print("OVERLAY_OK", len(mods), n_conv)
'''

STEP_SCRIPT = PRELUDE + r'''
# ---- the reference's whole detector against the replay harness ----------------------------------------------------------------------------
# FromVoxelToPoint (detectors/fv2p.py) built by the reference's own Detector3DTemplate.build_networks from the real fv2p.yaml (range halved,
# 2 048 key points, 1 024 -> 128 proposals, 32 RoIs, no dropout: the reduced step of tests/test_fv2p_step_gpu.py) over this repo's pcdet.ops,
# and the harness's FV2PDetector with the SAME parameters (the reference's state dict, two wrapper prefixes renamed, strict) on the SAME two
# clouds.  Both run a training forward on the host: C-ABI calls answered by the oracle, sparse convs by the oracle's conv shim.  The
# second stage is compared on the RoIs the reference's (random) sampler drew; the harness's own sampler is pinned separately (4).
import numpy as np, torch, yaml, os
torch.Tensor.cuda = lambda self, *a, **k: self
def ED(d):
    return ed.EasyDict({k: ED(v) for k, v in d.items()}) if isinstance(d, dict) else d
from pcdet.models.detectors.fv2p import FromVoxelToPoint
import oracle
from oracle.backend import oracle_backend
from oracle.spconv_cpu import cpu_mirror
from fv2p_harness import fv2p_model as fm, synth
from fv2p_harness.backbone import mean_vfe
cfg = ED(yaml.safe_load(open("/root/reference/tools/cfgs/kitti_models/FV2P/fv2p.yaml"))["MODEL"])
cfg.POST_PFE.NUM_KEYPOINTS = 2048
cfg.ROI_HEAD.NMS_CONFIG.TRAIN.NMS_PRE_MAXSIZE, cfg.ROI_HEAD.NMS_CONFIG.TRAIN.NMS_POST_MAXSIZE = 1024, 128
cfg.ROI_HEAD.TARGET_CONFIG.ROI_PER_IMAGE, cfg.ROI_HEAD.ROI_POINT_POOL.NUM_SAMPLED_POINTS, cfg.ROI_HEAD.DP_RATIO = 32, 128, 0

class Small(fm.FV2PConfig):
    point_cloud_range, grid_size = (0.0, -20.0, -3.0, 35.2, 20.0, 1.0), (704, 800, 40)
    num_keypoints, nms_pre, nms_post, roi_per_image, num_sampled_points, dp_ratio = 2048, 1024, 128, 32, 128, 0.0
rng = np.array(Small.point_cloud_range, np.float32)
dataset = types.SimpleNamespace(class_names=["Car"], point_feature_encoder=types.SimpleNamespace(num_point_features=4), grid_size=np.array(Small.grid_size),
                                point_cloud_range=rng, voxel_size=list(Small.voxel_size))
torch.manual_seed(0)
ref = FromVoxelToPoint(model_cfg=cfg, num_class=1, dataset=dataset)
assert [type(m).__name__ for m in ref.module_list] == ["MeanVFE", "VoxelResBackBone8x", "HeightCompression", "BaseBEVBackbone", "AnchorHeadSingle",
                                                        "ResidualVoxelToPointDecoder", "PointHeadSimple", "IoUGuidedRoIHead"]
own = fm.FV2PDetector(Small)
ren = lambda k: (k.replace("roi_head.CGE_module.corners_up_layer.", "roi_head.cge_up.").replace("roi_head.CGE_module.corners_inter_layer.", "roi_head.cge_inter.")
                  .replace("roi_head.feature_fusion.fuse_layer.", "roi_head.feature_fusion."))
own.load_state_dict({ren(k): v for k, v in ref.state_dict().items() if k != "global_step"}, strict=True)      # 404 entries, names and shapes
clouds, boxes, vox = [], [], []
for b in range(2):
    pts, bx = synth.lidar_cloud(40 + b, 4096, pc_range=rng, return_boxes=True)
    clouds.append(torch.from_numpy(pts)), boxes.append(bx), vox.append(oracle.points_to_voxel(pts, synth.KITTI_VOXEL, rng, 5, 16000))
gt = fm.pad_gt_boxes(boxes, "cpu")
voxels, nump = torch.from_numpy(np.concatenate([v for v, c, k in vox])), torch.from_numpy(np.concatenate([k for v, c, k in vox]))
coords = torch.from_numpy(np.concatenate([np.concatenate([np.full((c.shape[0], 1), b, np.int32), c], 1) for b, (v, c, k) in enumerate(vox)]))
points = torch.cat([torch.cat((torch.full((c.shape[0], 1), float(b)), c), 1) for b, c in enumerate(clouds)])
ref_cpu, own_cpu = cpu_mirror(ref).train(), cpu_mirror(own).train()
stash, layer = {}, ref_cpu.roi_head.proposal_layer
def proposal_layer(batch_dict, nms_config):
    out = layer(batch_dict, nms_config=nms_config)
    stash["rois"], stash["roi_scores"] = out["rois"].clone(), out["roi_scores"].clone()          # compared with the harness's proposals below
    # With random weights no proposal reaches the foreground thresholds and the regression / corner / IoU terms of the second stage
    # would be zero on both sides.  The head is therefore fed jittered ground-truth boxes from here on (a third nearly exact, a third
    # displaced by ~0.3 m, a third by metres), so the reference's sampler draws foreground, hard and easy background.
    gen = torch.Generator().manual_seed(9)
    for smp in range(out["rois"].shape[0]):
        boxes = batch_dict["gt_boxes"][smp]
        boxes = boxes[boxes[:, 3] > 0][:, :7]
        pick = boxes[torch.arange(out["rois"].shape[1]) % boxes.shape[0]]
        tier = (torch.arange(out["rois"].shape[1]) // boxes.shape[0]) % 3
        sigma = torch.tensor([0.03, 0.3, 3.0])[tier][:, None] * torch.tensor([1.0, 1.0, 0.3, 0.2, 0.2, 0.2, 0.3])[None]
        out["rois"][smp] = pick + sigma * torch.randn(pick.shape, generator=gen)
    return out
ref_cpu.roi_head.proposal_layer = proposal_layer
close = lambda a, b, tol: float((a - b).abs().max()) <= tol * max(1.0, float(b.abs().max()))
# The reference's RoI sampler draws from numpy's and torch's global generators (proposal_target_layer.py:97-147): seeded here, so that every
# run of this test compares the same 64 RoIs.  (Unseeded, one run in ~8 drew RoIs whose corner loss sits at a tie of its two orientations
# - torch.min over the flipped box, roi_withiou_head_template.py - where the reference's and the harness's roundings pick different branches:
# same loss to 1e-5, regression-layer gradients 1e-2 apart, and the bound below failed.)
np.random.seed(7)
torch.manual_seed(7)
with oracle_backend():
    bd = {"batch_size": 2, "points": points, "voxels": voxels, "voxel_num_points": nump, "voxel_coords": coords, "gt_boxes": gt}
    ret, tb, _ = ref_cpu(bd)
    fr = ref_cpu.roi_head.forward_ret_dict
    out, levels = own_cpu.backbone_3d(mean_vfe(voxels, nump), coords, 2)
    bev, loss_rpn, prop_scores, prop_boxes = own_cpu.dense_branch(out, 2, gt)
    key, feats = own_cpu.post_pfe(clouds, levels, None)
    loss_point, point_scores = own_cpu.point_head(key, feats, gt)
    rois, roi_scores = own_cpu.roi_head.proposals(prop_scores, prop_boxes)
    s_rois, s_gt, s_iou = fr["rois"].detach(), fr["gt_of_rois_src"].detach(), fr["gt_iou_of_rois"].detach()
    streams = own_cpu.roi_head.roi_streams(bev, s_rois)
    cls, reg = own_cpu.roi_head.predict(key, feats, point_scores, s_rois, streams)
    loss_rcnn = own_cpu.roi_head.losses(s_rois, s_gt, own_cpu.roi_head.canonical_targets(s_rois, s_gt), s_iou, cls, reg[:, 1:], reg[:, :1])
assert torch.equal(key.view(-1, 3), bd["point_coords"][:, 1:]), "key points"
assert close(feats, bd["point_features"], 1e-6), "decoded point features"
assert close(bev, bd["spatial_features_before_head"], 1e-6), "BEV features"
assert close(loss_rpn.detach(), tb["rpn_loss"] if torch.is_tensor(tb["rpn_loss"]) else torch.tensor(tb["rpn_loss"]), 1e-6), "first-stage loss"
assert abs(float(loss_point) - float(tb["point_loss_cls"])) <= 1e-6 * float(tb["point_loss_cls"]), "point loss"
assert close(rois, stash["rois"], 1e-6) and torch.equal(roi_scores, stash["roi_scores"]), "proposals"
assert close(cls.view(-1), fr["rcnn_cls"].view(-1).detach(), 1e-4) and close(reg[:, 1:], fr["rcnn_reg"].detach(), 1e-4) and close(reg[:, :1], fr["rcnn_iouscore"].detach(), 1e-4)
ref_rcnn = float(tb["rcnn_loss"]) + float(tb["rcnn_loss_iouscore"])      # get_loss logs 'rcnn_loss' before it adds the IoU-score term (roi_withiou_head_template.py:274-277)
assert abs(float(loss_rcnn) - ref_rcnn) <= 1e-5 * max(1.0, ref_rcnn), "second-stage loss"
assert tb["num_sample_fg"] > 0 and tb["num_sample_bg"] > 0 and float(tb["rcnn_loss_reg"]) > 0 and float(tb["rcnn_loss_iouscore"]) > 0, tb   # every term carries weight
total = float(loss_rpn + loss_point + loss_rcnn)
assert abs(total - float(ret["loss"])) <= 1e-6 * float(ret["loss"]), (total, float(ret["loss"]))
# ... and the same gradients: backward through the reference's graph (this package's autograd Functions: batched three_interpolate, grouping,
# BEV bilinear) and through the harness's (stacked variants, point-major MLPs, first set-abstraction layer per point), parameter by parameter
with oracle_backend():
    ret["loss"].backward()
    (loss_rpn + loss_point + loss_rcnn).backward()
ref_grads = {ren(k): p.grad for k, p in ref_cpu.named_parameters()}
own_grads = dict((k, p.grad) for k, p in own_cpu.named_parameters())
assert set(ref_grads) == set(own_grads) and all(g is not None for g in own_grads.values())
scale = max(float(g.norm()) for g in ref_grads.values())
# (a conv bias that feeds BatchNorm has a zero gradient up to rounding: the denominator has a floor of 1e-4 of the largest gradient norm)
ranked = sorted(((float((own_grads[k] - g).norm()) / max(float(g.norm()), 1e-4 * scale), k) for k, g in ref_grads.items()), reverse=True)
worst = ranked[0]
assert worst[0] < 3e-3, ranked[:5]      # relative L2 per parameter; the largest (1e-3) are the RoI head's point-stream biases: fp32 sums over 8 192 pooled rows in two layouts
print("STEP_OK", total, float(ret["loss"]), worst)
'''




def run_overlay(tmp_path, script):
    """A fresh interpreter whose `pcdet` is the reference's package with `ops` swapped for this repo's."""
    root = tmp_path / "pcdet"
    root.mkdir()
    for name in os.listdir(REF):
        if name != "ops":
            os.symlink(os.path.join(REF, name), root / name)
    os.symlink(os.path.join(REPO, "from-voxel-to-point_amd", "pcdet", "ops"), root / "ops")
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([str(tmp_path), os.path.join(REPO, "from-voxel-to-point_amd"), REPO]),
               FV2P_GOLDEN=os.path.join(REPO, "tests", "golden"))
    return subprocess.run([sys.executable, "-c", script], env=env, cwd=str(tmp_path), capture_output=True, text=True, timeout=900)


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not present (GPU box)")
def test_reference_models_import_against_this_ops_package(tmp_path):
    out = run_overlay(tmp_path, SCRIPT)
    assert out.returncode == 0 and "OVERLAY_OK" in out.stdout, out.stderr[-3000:]


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not present (GPU box)")
def test_reference_detector_and_harness_compute_the_same_training_step(tmp_path):
    """VERDICT r1 item 1's parity bar for the headline workload: the reference's own FromVoxelToPoint detector and the replay harness, same
    parameters, same clouds -> the same key points, features, proposals, head outputs and losses (see STEP_SCRIPT)."""
    out = run_overlay(tmp_path, STEP_SCRIPT)
    assert out.returncode == 0 and "STEP_OK" in out.stdout, out.stderr[-3000:]
