"""MdeformConvBlock — the block DCNBEVBackbone (pcdet/models/backbones_2d/dcn_bev_backbone.py:56-62) and the MGAF
head instantiate (reference modules/mdeformable_conv_block.py:32-96): a plain conv predicts (offset_h, offset_w, mask)
per kernel tap and deformable group, then a bias-free ModulatedDeformConv adapts the features."""
import torch
from torch import nn

from .modulated_deform_conv import ModulatedDeformConv


class MdeformConvBlock(nn.Module):

    def __init__(self, in_channels, out_channels, kernel_size=3, deformable_groups=4):
        super(MdeformConvBlock, self).__init__()
        offset_mask_channels = kernel_size * kernel_size * (2 + 1)
        self.conv_offset_mask = nn.Conv2d(in_channels, deformable_groups * offset_mask_channels, kernel_size=kernel_size, stride=1,
                                          padding=(kernel_size - 1) // 2, bias=True)
        self.conv_adaption = ModulatedDeformConv(in_channels, out_channels, stride=1, kernel_size=kernel_size,
                                                 padding=(kernel_size - 1) // 2, deformable_groups=deformable_groups, bias=False)
        self.init_offset()

    def init_offset(self):
        self.conv_offset_mask.weight.data.zero_()
        self.conv_offset_mask.bias.data.zero_()

    def init_weights(self):
        pass

    def forward(self, x):
        offset_mask = self.conv_offset_mask(x)
        o1, o2, mask = torch.chunk(offset_mask, 3, dim=1)
        offset = torch.cat((o1, o2), dim=1)
        mask = torch.sigmoid(mask)
        return self.conv_adaption(x, offset, mask)
