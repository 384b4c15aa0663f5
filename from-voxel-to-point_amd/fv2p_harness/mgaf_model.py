"""Layer replay of the MGAF-3DSSD detector's dense part (BASELINE configs[3]: mgaf-3dssd_3classes.yaml) as a consumer of this
repo's `pcdet.ops`: VoxelResBackBone8x -> HeightCompression -> DCNBEVBackbone (three levels, one MdeformConvBlock per level,
pcdet/models/backbones_2d/dcn_bev_backbone.py:10-132) -> CenterAFHeadSingle (shared conv, modulated deformable feature
adaption with four deformable groups, seven convolutional heads with the segmentation-guided attention,
pcdet/models/dense_heads/center_af_head_single.py:8-110).

What is replayed is every layer and therefore every kernel of the forward and backward pass; the head's target assignment
and its seven loss terms (center_af_head_template.py, ~600 lines of torch glue: Gaussian heat-maps, gathered L1 terms, bin
losses) are NOT — the loss here is a fixed surrogate (mean square of every head output) that sends a gradient through every
layer.  The numbers of this workload therefore price the ops (DCNv2 forward / backward at MGAF shapes beside the sparse
backbone), not the reference's loss bookkeeping."""
from functools import partial

import torch
import torch.nn as nn

from pcdet.ops.DeformableConvolutionV2PyTorch.modules.mdeformable_conv_block import MdeformConvBlock
from pcdet.ops.DeformableConvolutionV2PyTorch.modules.modulated_deform_conv import ModulatedDeformConv

from .backbone import VoxelResBackBone8x


class MGAFConfig:
    grid_size = (1408, 1600, 40)
    num_point_features = 4
    layer_nums, layer_strides, num_filters = (5, 5, 5), (1, 2, 2), (128, 256, 256)
    upsample_strides, num_upsample_filters = (1, 2, 4), (256, 256, 256)
    shared_fc = (256,)
    head_deformable_groups = 4
    heads = (("hm", 3), ("offset", 2), ("height", 1), ("dim", 3), ("rot", 24), ("segm", 1), ("iouscore", 1))   # 3 classes
    head_conv = 128


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
        for blk, de in zip(self.blocks, self.deblocks):
            x = blk(x)
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


class MGAFDetector(nn.Module):
    def __init__(self, cfg=MGAFConfig, offset_init_std=0.05):
        super().__init__()
        self.cfg = cfg
        self.backbone_3d = VoxelResBackBone8x(cfg.num_point_features, list(cfg.grid_size))     # BACKBONE_3D of both MGAF yamls
        self.backbone_2d = DCNBEVBackbone(cfg, 256)
        self.dense_head = CenterAFHead(cfg, self.backbone_2d.num_bev_features)
        # the reference zero-initialises the offset / mask predictors (all offsets 0 at step 0); a trained net has moved away from
        # that, so the replay starts them at small random values: sampling positions are fractional, as in any later step
        if offset_init_std > 0:
            for m in self.modules():
                if isinstance(m, (MdeformConvBlock, FeatureAdaption)):
                    nn.init.normal_(m.conv_offset_mask.weight, std=offset_init_std)

    def forward(self, voxel_features, voxel_coords, batch_size):
        out, _ = self.backbone_3d(voxel_features, voxel_coords, batch_size)
        dense = out.dense()
        spatial = dense.view(batch_size, dense.shape[1] * dense.shape[2], dense.shape[3], dense.shape[4])
        preds = self.dense_head(self.backbone_2d(spatial))
        return sum(p.square().mean() for p in preds.values())
