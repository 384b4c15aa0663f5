"""Replay of the MGAF-3DSSD detector (BASELINE configs[3]: mgaf-3dssd_3classes.yaml) as a consumer of this repo's `pcdet.ops`:
VoxelResBackBone8x -> HeightCompression -> DCNBEVBackbone (three levels, one MdeformConvBlock per level,
pcdet/models/backbones_2d/dcn_bev_backbone.py:10-132) -> CenterAFHeadSingle (shared conv, modulated deformable feature
adaption with four deformable groups, seven convolutional heads with the segmentation-guided attention,
pcdet/models/dense_heads/center_af_head_single.py:8-148) -> the head's target assignment (CenterTargetAssigner,
keypoint_assigner/centertarget_assigner.py:24-209) and its eight loss terms (center_af_head_template.py:193-460).

Target assignment: the reference draws the maps on the HOST (numpy + cv2, one sample at a time, gt boxes copied to the host and
the maps back: centertarget_assigner.py:41, 72-82); `center_targets` states the same rule as batch tensor ops on the boxes'
device — no host round trip.  Gaussian heat-map, indices, masks and regression targets follow the reference's float64 / float32
arithmetic operation for operation (pinned by tests/golden/pyref_center_targets.npz, written from the reference class); the
segmentation map is cv2.fillConvexPoly in the reference, a library this image lacks: it is restated as "pixel centre inside the
closed quadrilateral, or on one of its 8-connected edge lines" (OpenCV falls back from LINE_AA to 8-connected lines on non-8-bit
images) and is the one target that is NOT pinned.  The per-pixel height map the assigner also returns feeds no loss
(center_af_head_template.py:263-278 reads the per-object target) and is not produced."""
from functools import partial

import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from pcdet.ops.DeformableConvolutionV2PyTorch.modules.mdeformable_conv_block import MdeformConvBlock
from pcdet.ops.DeformableConvolutionV2PyTorch.modules.modulated_deform_conv import ModulatedDeformConv

from pcdet.ops.iou3d_nms import iou3d_nms_utils

from .backbone import VoxelResBackBone8x
from .fv2p_model import box_corners, sigmoid_focal


class MGAFConfig:
    grid_size = (1408, 1600, 40)
    num_point_features = 4
    layer_nums, layer_strides, num_filters = (5, 5, 5), (1, 2, 2), (128, 256, 256)
    upsample_strides, num_upsample_filters = (1, 2, 4), (256, 256, 256)
    shared_fc = (256,)
    head_deformable_groups = 4
    heads = (("hm", 3), ("offset", 2), ("height", 1), ("dim", 3), ("rot", 24), ("segm", 1), ("iouscore", 1))   # 3 classes
    head_conv = 128
    # TARGET_ASSIGNER_CONFIG / LOSS_CONFIG of mgaf-3dssd_3classes.yaml:134-175
    point_cloud_range = (0.0, -40.0, -3.0, 70.4, 40.0, 1.0)
    voxel_size = (0.05, 0.05, 0.1)
    feature_map_stride, gaussian_minoverlap, gaussian_minradius, max_objs = 8, 0.01, 2, 50
    num_iouscore_training_samples = 24
    loss_weights = dict(hm=1.0, offset=1.0, height=1.0, dim=1.0, rot=1.0, segm=1.0, corner=1.0, iouscore=1.0)
    rot_bins, iou_fg_thresh, iou_bg_thresh = 12, 0.75, 0.25


class DCNBEVBackbone(nn.Module):
    def __init__(self, cfg, cin):
        super().__init__()
        bn = partial(nn.BatchNorm2d, eps=1e-3, momentum=0.01)
        self.blocks, self.deblocks = nn.ModuleList(), nn.ModuleList()
        for n, s, f, us, uf in zip(cfg.layer_nums, cfg.layer_strides, cfg.num_filters, cfg.upsample_strides, cfg.num_upsample_filters):
            seq = [nn.ZeroPad2d(1), nn.Conv2d(cin, f, 3, stride=s, padding=0, bias=False), bn(f), nn.ReLU()]
            for _ in range(n):
                seq += [nn.Conv2d(f, f, 3, padding=1, bias=False), bn(f), nn.ReLU()]
            self.blocks.append(nn.Sequential(*seq))
            self.deblocks.append(nn.Sequential(MdeformConvBlock(f, f, deformable_groups=1), bn(f), nn.ReLU(),
                                               nn.ConvTranspose2d(f, uf, us, stride=us, bias=False), bn(uf), nn.ReLU()))
            cin = f
        self.num_bev_features = sum(cfg.num_upsample_filters)

    def forward(self, x):
        ups = []
        from .fv2p_model import BEVBackbone
        for blk, de in zip(self.blocks, self.deblocks):
            x = BEVBackbone._block(blk, x)   # the leading ZeroPad2d + Conv2d pair as one zero-padded convolution (same sums, no padded copy)
            ups.append(de(x))
        return torch.cat(ups, dim=1)


class FeatureAdaption(nn.Module):
    """feature_adaptor/mdeformable_convs.py:14-78: offsets and masks from a plain conv, DCNv2, ReLU."""

    def __init__(self, channels, deformable_groups):
        super().__init__()
        self.conv_offset_mask = nn.Conv2d(channels, deformable_groups * 27, 3, padding=1, bias=True)
        self.conv_adaption = ModulatedDeformConv(channels, channels, stride=1, kernel_size=3, padding=1, deformable_groups=deformable_groups, bias=False)

    def forward(self, x):
        o1, o2, mask = torch.chunk(self.conv_offset_mask(x), 3, dim=1)
        return torch.relu(self.conv_adaption(x, torch.cat((o1, o2), dim=1), torch.sigmoid(mask)))


class CenterAFHead(nn.Module):
    def __init__(self, cfg, cin):
        super().__init__()
        layers, pre = [], cin
        for f in cfg.shared_fc:
            layers += [nn.Conv2d(pre, f, 3, padding=1, bias=False), nn.BatchNorm2d(f), nn.ReLU()]
            pre = f
        self.shared_conv_layer = nn.Sequential(*layers)
        self.feature_adapt = FeatureAdaption(pre, cfg.head_deformable_groups)
        self.heads = nn.ModuleDict({name: nn.Sequential(nn.Conv2d(pre, cfg.head_conv, 3, padding=1, bias=False),
                                                        nn.BatchNorm2d(cfg.head_conv, eps=1e-3, momentum=0.01), nn.ReLU(),
                                                        nn.Conv2d(cfg.head_conv, out, 1, bias=True)) for name, out in cfg.heads})

    def forward(self, x):
        x = self.feature_adapt(self.shared_conv_layer(x))
        segm = self.heads["segm"](x)
        att = x + torch.sigmoid(segm.detach()).expand_as(x) * x       # mask-guided attention (center_af_head_single.py:84-92)
        preds = {"segm": segm}
        for name, head in self.heads.items():
            if name != "segm":
                preds[name] = head(att)
        return preds


# ---------------------------------------------------------------- target assignment ---------------------------------------------------
def gaussian_radius(h, w, min_overlap):
    """center_utils.gaussian_radius (:98-122) on float64 tensors."""
    b1 = h + w
    c1 = w * h * (1 - min_overlap) / (1 + min_overlap)
    r1 = (b1 - torch.sqrt(b1 ** 2 - 4 * c1)) / 2
    b2 = 2 * (h + w)
    c2 = (1 - min_overlap) * w * h
    r2 = (b2 - torch.sqrt(b2 ** 2 - 16 * c2)) / 8
    a3 = 4 * min_overlap
    b3 = -2 * min_overlap * (h + w)
    c3 = (min_overlap - 1) * w * h
    r3 = (b3 + torch.sqrt(b3 ** 2 - 4 * a3 * c3)) / (2 * a3)
    return torch.minimum(torch.minimum(r1, r2), r3)


@torch.no_grad()
def center_targets(gt, cfg, num_classes):
    """CenterTargetAssigner.assign_targets (centertarget_assigner.py:24-209) for the batch, on gt's device.
    gt (B, M, 8) zero padded [x, y, z, dx, dy, dz, heading, class] -> dict with the reference's keys:
    hm_target (B, C, H, W), anno_box_target (B, K, 7), ind_target (B, K) int64, mask_target (B, K) uint8, segm_target (B, 1, H, W),
    src_box_target (B, K, 7), xsys_target (B, K, 2), batch_gtboxes_src (B, M, 8).  K = MAX_OBJS; object k of a sample is its k-th box."""
    dev, b, m = gt.device, gt.shape[0], gt.shape[1]
    k = cfg.max_objs
    lo = torch.tensor(cfg.point_cloud_range[:3], dtype=torch.float32, device=dev)
    hi = torch.tensor(cfg.point_cloud_range[3:], dtype=torch.float32, device=dev)
    vs = torch.tensor(cfg.voxel_size, dtype=torch.float64, device=dev)
    stride = cfg.feature_map_stride
    sy = int(round((cfg.point_cloud_range[4] - cfg.point_cloud_range[1]) / cfg.voxel_size[1] / stride))
    sx = int(round((cfg.point_cloud_range[3] - cfg.point_cloud_range[0]) / cfg.voxel_size[0] / stride))
    boxes = gt[:, :, :7]
    # the reference cuts the trailing all-zero rows (rows whose seven numbers SUM to zero, :50-53) and takes the first K of the rest
    nonzero = boxes.sum(-1) != 0
    pos = torch.arange(m, device=dev)
    last = torch.where(nonzero, pos, torch.zeros_like(pos)).amax(1)                       # cnt of :50-52 (row 0 always stays)
    kk = min(k, m)
    bx = boxes[:, :kk]
    live = (pos[:kk][None] <= last[:, None])
    cls = gt[:, :kk, 7].to(torch.int8).long() - 1                                        # astype(np.int8) then - 1 (:56, 126)
    dimx = bx[..., 3].double() / vs[0] / stride                                          # float32 box / float64 voxel size (:127-128)
    dimy = bx[..., 4].double() / vs[1] / stride
    ok = live & (dimx > 0) & (dimy > 0)
    radius = gaussian_radius(torch.ceil(dimx), torch.ceil(dimy), cfg.gaussian_minoverlap)
    radius = torch.where(ok, radius, torch.zeros_like(radius)).nan_to_num(0.0).clamp(min=0).long().clamp(min=cfg.gaussian_minradius)   # max(int(r), minradius)
    cx = (bx[..., 0] - lo[0]).double() / vs[0] / stride                                  # float32 subtraction, float64 division (:138-139)
    cy = (bx[..., 1] - lo[1]).double() / vs[1] / stride
    cxi, cyi = torch.round(cx).long(), torch.round(cy).long()                            # np.around: half to even, as torch.round
    ok = ok & (cxi >= 0) & (cxi < sx) & (cyi >= 0) & (cyi < sy)
    # heat map: element-wise maximum of the objects' Gaussians, float64 values rounded to float32 once (:160, center_utils.py:139-155)
    ys = torch.arange(sy, device=dev).view(1, 1, sy, 1)
    xs = torch.arange(sx, device=dev).view(1, 1, 1, sx)
    dy, dx = ys - cyi[..., None, None], xs - cxi[..., None, None]
    r = radius[..., None, None]
    sigma = (2 * radius + 1).double()[..., None, None] / 6
    g = torch.exp(-(dx * dx + dy * dy).double() / (2 * sigma * sigma))
    g = torch.where(g < torch.finfo(torch.float64).eps, torch.zeros_like(g), g)          # h[h < eps * h.max()] = 0, h.max() = 1
    g = torch.where((dx.abs() <= r) & (dy.abs() <= r) & ok[..., None, None], g, torch.zeros_like(g)).float()
    hm = gt.new_zeros(b, num_classes, sy, sx)
    for c in range(num_classes):
        hm[:, c] = torch.where((cls == c)[..., None, None], g, torch.zeros_like(g)).amax(1)
    okf = ok.unsqueeze(-1)
    ind = torch.where(ok, sx * cyi + cxi, torch.zeros_like(cxi))
    anno = torch.cat(((cx - cxi).float().unsqueeze(-1), (cy - cyi).float().unsqueeze(-1), bx[..., 2:7]), -1) * okf
    xsys = torch.stack((cxi, cyi), -1).float() * okf
    src = bx * okf
    # segmentation map: the BEV quadrilateral of every live object (corners 0..3 of boxes_to_corners_3d, clipped to the range, in
    # map pixels, rounded to integers :181-190), filled as cv2.fillConvexPoly(lineType 8) does — see the module docstring
    corners = box_corners(bx.reshape(-1, 7)).view(b, kk, 8, 3)[:, :, :4, :2]
    qx = torch.round((corners[..., 0].clamp(lo[0], hi[0]) - lo[0]).double() / vs[0] / stride).long()     # (B, K, 4)
    qy = torch.round((corners[..., 1].clamp(lo[1], hi[1]) - lo[1]).double() / vs[1] / stride).long()
    px, py = xs.view(1, 1, 1, sx), ys.view(1, 1, sy, 1)
    inside_pos = torch.ones(b, kk, sy, sx, dtype=torch.bool, device=dev)
    inside_neg = torch.ones_like(inside_pos)
    on_edge = torch.zeros_like(inside_pos)
    for e in range(4):
        x0, y0 = qx[..., e, None, None], qy[..., e, None, None]
        x1, y1 = qx[..., (e + 1) % 4, None, None], qy[..., (e + 1) % 4, None, None]
        ex, ey = x1 - x0, y1 - y0
        cross = ex * (py - y0) - ey * (px - x0)
        inside_pos &= cross >= 0
        inside_neg &= cross <= 0
        # 8-connected line: one pixel per step of the major axis, the minor coordinate rounded to the nearest row / column
        xmajor = ex.abs() >= ey.abs()
        in_x = (px >= torch.minimum(x0, x1)) & (px <= torch.maximum(x0, x1))
        in_y = (py >= torch.minimum(y0, y1)) & (py <= torch.maximum(y0, y1))
        near_x = (2 * (cross.abs()) <= ex.abs()) & in_x & (ex != 0)       # |py - yline(px)| <= 1/2  <=>  2 |cross| <= |ex|
        near_y = (2 * (cross.abs()) <= ey.abs()) & in_y & (ey != 0)
        point = (ex == 0) & (ey == 0) & (px == x0) & (py == y0)
        on_edge |= torch.where(xmajor, near_x, near_y) | point
    segm = (((inside_pos | inside_neg | on_edge) & ok[..., None, None]).any(1, keepdim=True)).float()
    pad = (lambda t: F.pad(t, (0, 0, 0, k - kk)) if t.dim() == 3 else F.pad(t, (0, k - kk))) if kk < k else (lambda t: t)   # K slots always
    return {"hm_target": hm, "anno_box_target": pad(anno), "ind_target": pad(ind), "mask_target": pad(ok.to(torch.uint8)), "segm_target": segm,
            "src_box_target": pad(src), "xsys_target": pad(xsys), "batch_gtboxes_src": gt.clone()}


# ---------------------------------------------------------------- losses ---------------------------------------------------------------
def gather_map(feat, ind):
    """center_utils._transpose_and_gather_feat (:250-259): (B, C, H, W), (B, K) -> (B, K, C)."""
    b, c = feat.shape[:2]
    return feat.view(b, c, -1).gather(2, ind.unsqueeze(1).expand(-1, c, -1)).transpose(1, 2)


def decode_rot_binres(reg, bins):
    """box_utils.decode_rot_binres (:366-406): (N, 2 bins) -> (N, 1) heading in (-pi, pi]."""
    per = 2 * math.pi / bins
    which = reg[:, :bins].argmax(1)
    res = reg[:, bins:].gather(1, which.unsqueeze(1)).squeeze(1) * (per / 2)
    ry = (which.float() * per + res) % (2 * math.pi)
    ry = torch.where(ry > math.pi, ry - 2 * math.pi, ry)
    return ry.view(-1, 1)


def center_losses(preds, tg, cfg, peaks=None):
    """CenterAFHeadTemplate.get_loss (center_af_head_template.py:193-460) on the seven head maps `preds` and the targets `tg`
    -> (total, dict of the eight weighted terms).  Tensor ops only; the one op of this repo it reaches is boxes_iou3d_gpu.
    `peaks` [B, K] (parity tests): heat-map cells to score in the IoU-score term instead of this run's own top K, so that two float
    implementations are compared on the same cells (two near-equal peaks may swap between them); the term records its cells in
    out["_iou_peaks"]."""
    w = cfg.loss_weights
    b = preds["hm"].shape[0]
    mask, ind = tg["mask_target"].bool(), tg["ind_target"]
    anno = tg["anno_box_target"]
    out = {}
    # heat map: CenterNetFocalLoss (loss_utils.py:410-462)
    p = torch.sigmoid(preds["hm"]).clamp(1e-4, 1 - 1e-4)
    t = tg["hm_target"]
    pos, neg = t.eq(1).float(), t.lt(1).float()
    pos_loss = (torch.log(p) * (1 - p).pow(2) * pos).sum()
    neg_loss = (torch.log(1 - p) * p.pow(2) * (1 - t).pow(4) * neg).sum()
    npos = pos.sum()
    out["hm"] = w["hm"] * torch.where(npos == 0, -neg_loss, -(pos_loss + neg_loss) / npos.clamp_min(1.0))
    # gathered L1 terms: CenterNetResLoss 'l1' (loss_utils.py:465-509): mean over the objects, sum over the channels
    any_obj = mask.any()

    def res(name, target):
        pred = gather_map(preds[name], ind)[mask]
        return F.l1_loss(pred, target[mask], reduction="none").mean(0).sum() if any_obj else pred.sum() * 0.0
    out["offset"] = w["offset"] * res("offset", anno[:, :, 0:2])
    out["height"] = w["height"] * res("height", anno[:, :, 2:3])
    out["dim"] = w["dim"] * res("dim", anno[:, :, 3:6])
    # heading: bin cross entropy + smooth-L1 of the normalised residual in the labelled bin (loss_utils.get_rot_binres_loss :334-404)
    bins = cfg.rot_bins
    rot_pred = gather_map(preds["rot"], ind)[mask]
    if any_obj:
        per = 2 * math.pi / bins
        heading = anno[:, :, 6][mask] % (2 * math.pi)
        shift = (heading + per / 2) % (2 * math.pi)
        bin_label = (shift / per).floor().long()
        res_label = (shift - (bin_label.float() * per + per / 2)) / (per / 2)
        onehot = F.one_hot(bin_label, bins).float()
        out["rot"] = w["rot"] * (F.cross_entropy(rot_pred[:, :bins], bin_label) +
                                 F.smooth_l1_loss((rot_pred[:, bins:] * onehot).sum(1), res_label))
    else:
        out["rot"] = rot_pred.sum() * 0.0
    # segmentation: sigmoid focal loss, every pixel weighted 1 / positives of its sample (:314-342)
    sp = preds["segm"].permute(0, 2, 3, 1).reshape(b, -1, 1)
    st = tg["segm_target"].permute(0, 2, 3, 1).reshape(b, -1, 1)
    weights = ((st == 0).float() + (st > 0).float()).squeeze(-1) / (st > 0).sum(1).float().clamp(min=1.0)
    out["segm"] = w["segm"] * sigmoid_focal(sp, st, weights).sum() / b
    # corner loss on the boxes decoded at the ground-truth centres (:344-365, 464-516; loss_utils.get_corner_loss_mse :246-281)
    k = ind.shape[1]
    off = gather_map(preds["offset"], ind)
    xs = (tg["xsys_target"][:, :, 0:1] + off[:, :, 0:1]) * cfg.feature_map_stride * cfg.voxel_size[0] + cfg.point_cloud_range[0]
    ys = (tg["xsys_target"][:, :, 1:2] + off[:, :, 1:2]) * cfg.feature_map_stride * cfg.voxel_size[1] + cfg.point_cloud_range[1]
    rot = decode_rot_binres(gather_map(preds["rot"], ind).reshape(b * k, -1), bins).view(b, k, 1)
    decoded = torch.cat((xs, ys, gather_map(preds["height"], ind), gather_map(preds["dim"], ind), rot), 2)
    if any_obj:
        pc, gc = box_corners(decoded[mask]), box_corners(tg["src_box_target"][mask])
        corner = sum(F.mse_loss(pc[:, :, a], gc[:, :, a]) for a in range(3))
    else:
        corner = decoded.sum() * 0.0
    out["corner"] = w["corner"] * corner / 3.0
    # IoU score: the K best peaks of the max-pooled heat map decoded to boxes, labelled by their best 3-D IoU with a box of the
    # predicted class (:367-460, 518-598; loss_utils.get_iouscore_loss_bce :284-331)
    with torch.no_grad():
        kq = cfg.num_iouscore_training_samples
        hm = preds["hm"].detach()
        heat = hm * (F.max_pool2d(hm, 3, stride=1, padding=1) == hm).float()                  # center_utils._nms
        nc, sy, sx = heat.shape[1:]
        top_s, top_i = torch.topk(heat.view(b, nc, -1), kq)                                     # center_utils._topk
        top_s2, top_j = torch.topk(top_s.view(b, -1), kq)
        inds = top_i.view(b, -1).gather(1, top_j) if peaks is None else peaks
        pxs, pys = (inds % sx).float().unsqueeze(-1), torch.div(inds, sx, rounding_mode="floor").float().unsqueeze(-1)
        offq = gather_map(preds["offset"].detach(), inds)
        bxs = (pxs + offq[:, :, 0:1]) * cfg.feature_map_stride * cfg.voxel_size[0] + cfg.point_cloud_range[0]
        bys = (pys + offq[:, :, 1:2]) * cfg.feature_map_stride * cfg.voxel_size[1] + cfg.point_cloud_range[1]
        rotq = decode_rot_binres(gather_map(preds["rot"].detach(), inds).reshape(b * kq, -1), bins).view(b, kq, 1)
        qboxes = torch.cat((bxs, bys, gather_map(preds["height"].detach(), inds), gather_map(preds["dim"].detach(), inds), rotq), 2)
        qcls = gather_map(heat, inds).argmax(-1) + 1                                             # (B, K) predicted class id
        gtb = tg["batch_gtboxes_src"]
        ious = []
        for i in range(b):     # per sample as the reference (boxes_iou3d_gpu is a pairwise op on one sample's boxes)
            valid = gtb[i, :, :7].sum(1) != 0
            g7, gl = gtb[i, valid, :7].contiguous(), gtb[i, valid, 7].long()
            best = qboxes.new_zeros(kq)
            if g7.shape[0] > 0:
                iou = iou3d_nms_utils.boxes_iou3d_gpu(qboxes[i].contiguous(), g7)               # (K, G)
                same = qcls[i][:, None] == gl[None, :]
                best = torch.where(same, iou, torch.zeros_like(iou)).amax(1)                   # get_max_iou_with_same_class
                best = torch.where(same.any(1), best, torch.zeros_like(best))
            ious.append(best)
        iou_gt = torch.stack(ious).view(-1)
        fg, bg = iou_gt > cfg.iou_fg_thresh, iou_gt < cfg.iou_bg_thresh
        label = torch.where(fg, torch.ones_like(iou_gt), torch.where(bg, torch.zeros_like(iou_gt),
                                                                      (iou_gt - cfg.iou_bg_thresh) / (cfg.iou_fg_thresh - cfg.iou_bg_thresh)))
    score = gather_map(preds["iouscore"], inds).reshape(-1)
    bce = F.binary_cross_entropy(torch.sigmoid(score), label, reduction="none")
    out["iouscore"] = w["iouscore"] * bce.sum() / max(bce.numel(), 1)
    total = sum(out.values())
    out["_iou_peaks"] = inds
    return total, out


class MGAFDetector(nn.Module):
    def __init__(self, cfg=MGAFConfig, offset_init_std=0.05):
        super().__init__()
        self.cfg = cfg
        self.backbone_3d = VoxelResBackBone8x(cfg.num_point_features, list(cfg.grid_size))     # BACKBONE_3D of both MGAF yamls
        self.backbone_2d = DCNBEVBackbone(cfg, 256)
        self.dense_head = CenterAFHead(cfg, self.backbone_2d.num_bev_features)
        self.taps = None    # set to a dict to collect head maps, targets and loss terms (parity tests)
        self.iou_peaks = None   # parity tests: the heat-map cells the IoU-score term scores ([B, K]; None = this run's own top K)
        # the reference zero-initialises the offset / mask predictors (all offsets 0 at step 0); a trained net has moved away from
        # that, so the replay starts them at small random values: sampling positions are fractional, as in any later step
        if offset_init_std > 0:
            for m in self.modules():
                if isinstance(m, (MdeformConvBlock, FeatureAdaption)):
                    nn.init.normal_(m.conv_offset_mask.weight, std=offset_init_std)

    def forward(self, voxel_features, voxel_coords, batch_size, gt_boxes=None):
        """-> the training loss of MGAF3DSSD.get_training_loss (detectors/mgaf_3dssd.py:24-34: the dense head's eight terms; the yaml has
        no point head).  Without gt_boxes: the surrogate of round 2 (mean square of every head map), a gradient through every layer."""
        out, _ = self.backbone_3d(voxel_features, voxel_coords, batch_size)
        dense = out.dense()
        spatial = dense.view(batch_size, dense.shape[1] * dense.shape[2], dense.shape[3], dense.shape[4])
        if getattr(self, "bev_channels_last", False):   # bench.py --bev-channels-last: the 2-D part in NHWC (the DCN kernels' layout)
            spatial = spatial.contiguous(memory_format=torch.channels_last)
        preds = self.dense_head(self.backbone_2d(spatial))
        if gt_boxes is None:
            return sum(p.square().mean() for p in preds.values())
        targets = center_targets(gt_boxes, self.cfg, dict(self.cfg.heads)["hm"])
        loss, terms = center_losses(preds, targets, self.cfg, peaks=self.iou_peaks)
        if self.taps is not None:
            self.taps.update(preds=preds, targets=targets, terms=terms)
        return loss
