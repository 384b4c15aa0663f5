"""DeformRoIPooling / DeformRoIPoolingPack — module surface of the reference's modules/deform_psroi_pooling.py:13-130."""
import torch
from torch import nn

from ..functions.deform_psroi_pooling_func import DeformRoIPoolingFunction

_DeformRoIPooling = DeformRoIPoolingFunction.apply


class DeformRoIPooling(nn.Module):
    """forward(input [B, C, H, W], rois [R, 5], offset [R, 2 * classes, part, part]) -> [R, output_dim, pooled, pooled]."""

    def __init__(self, spatial_scale, pooled_size, output_dim, no_trans, group_size=1, part_size=None, sample_per_part=4, trans_std=.0):
        super().__init__()
        self.spatial_scale, self.pooled_size, self.output_dim, self.no_trans = spatial_scale, pooled_size, output_dim, no_trans
        self.group_size, self.part_size = group_size, pooled_size if part_size is None else part_size
        self.sample_per_part, self.trans_std = sample_per_part, trans_std

    def _pool(self, input, rois, offset, no_trans):
        return _DeformRoIPooling(input, rois, offset, self.spatial_scale, self.pooled_size, self.output_dim, no_trans, self.group_size,
                                 self.part_size, self.sample_per_part, self.trans_std)

    def forward(self, input, rois, offset):
        assert input.shape[1] == self.output_dim
        if self.no_trans:
            offset = input.new()
        return self._pool(input, rois, offset, self.no_trans)


class DeformRoIPoolingPack(DeformRoIPooling):
    """The pooling that learns its own shifts and a modulation mask (DCNv2): a plain pass, three Linear layers on it
    (the last one zero-initialised) -> 2 shift planes + 1 mask plane, then the shifted pass times sigmoid(mask)."""

    def __init__(self, spatial_scale, pooled_size, output_dim, no_trans, group_size=1, part_size=None, sample_per_part=4, trans_std=.0,
                 deform_fc_dim=1024):
        super().__init__(spatial_scale, pooled_size, output_dim, no_trans, group_size, part_size, sample_per_part, trans_std)
        self.deform_fc_dim = deform_fc_dim
        if not no_trans:
            bins = self.pooled_size * self.pooled_size
            self.offset_mask_fc = nn.Sequential(nn.Linear(bins * self.output_dim, deform_fc_dim), nn.ReLU(inplace=True),
                                                nn.Linear(deform_fc_dim, deform_fc_dim), nn.ReLU(inplace=True),
                                                nn.Linear(deform_fc_dim, bins * 3))
            nn.init.zeros_(self.offset_mask_fc[4].weight)
            nn.init.zeros_(self.offset_mask_fc[4].bias)

    def forward(self, input, rois):
        if self.no_trans:
            return self._pool(input, rois, input.new(), True)
        n = rois.shape[0]
        plain = self._pool(input, rois, input.new(), True)
        o1, o2, mask = torch.chunk(self.offset_mask_fc(plain.view(n, -1)).view(n, 3, self.pooled_size, self.pooled_size), 3, dim=1)
        return self._pool(input, rois, torch.cat((o1, o2), dim=1), False) * torch.sigmoid(mask)
