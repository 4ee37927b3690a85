"""DeepLabV3 decoder of the reference (GLfusion/models/deeplabv3.py:102-166) on the HIP engine.

ASPP specifics of this build: the five branch outputs are never concatenated -- the 1280->256
projection is evaluated as five accumulated K=256 contractions (ops.conv1x1_cat); dilated
taps that fall into the zero padding for a whole tile are skipped inside the conv kernel
(rate 36 on a 28x28 map reduces to its centre tap); the pooled branch is a per-frame row
broadcast."""
from __future__ import annotations

from typing import List

import torch
from torch import nn

from .. import ops
from .layers import AdaptiveAvgPool2d, BatchNorm2d, Conv2d, Dropout, ReLU, conv_bn_act
from ._utils import _SimpleSegmentationModel_iekd

__all__ = ["DeepLabV3_iekd", "DeepLabHead", "ASPP", "ASPPConv", "ASPPPooling"]


class DeepLabV3_iekd(_SimpleSegmentationModel_iekd):
    """deeplabv3.py:31-46."""
    pass


class ASPPConv(nn.Sequential):
    def __init__(self, in_channels: int, out_channels: int, dilation: int) -> None:
        super().__init__(Conv2d(in_channels, out_channels, 3, padding=dilation, dilation=dilation, bias=False),
                         BatchNorm2d(out_channels), ReLU())

    def forward_nhwc(self, x):
        return conv_bn_act(x, self[0], self[1], relu=True)

    def forward(self, x):
        return ops.from_nhwc(self.forward_nhwc(ops.to_nhwc(x)))


class ASPPPooling(nn.Sequential):
    def __init__(self, in_channels: int, out_channels: int) -> None:
        super().__init__(AdaptiveAvgPool2d(1), Conv2d(in_channels, out_channels, 1, bias=False),
                         BatchNorm2d(out_channels), ReLU())

    def forward_nhwc(self, x):
        h, w = x.shape[1], x.shape[2]
        p = ops.global_avgpool(x)                               # [N,1,1,C]
        p = conv_bn_act(p, self[1], self[2], relu=True)         # BN statistics over the N frames
        return ops.broadcast_hw(p, h, w)                        # bilinear from 1x1 == broadcast (deeplabv3.py:135)

    def forward(self, x):
        return ops.from_nhwc(self.forward_nhwc(ops.to_nhwc(x)))


class _Branch1x1(nn.Sequential):
    def forward_nhwc(self, x):
        return conv_bn_act(x, self[0], self[1], relu=True)

    def forward(self, x):
        return ops.from_nhwc(self.forward_nhwc(ops.to_nhwc(x)))


class ASPP(nn.Module):
    def __init__(self, in_channels: int, atrous_rates: List[int], out_channels: int = 256) -> None:
        super().__init__()
        modules: List[nn.Module] = [_Branch1x1(Conv2d(in_channels, out_channels, 1, bias=False),
                                               BatchNorm2d(out_channels), ReLU())]
        for rate in tuple(atrous_rates):
            modules.append(ASPPConv(in_channels, out_channels, rate))
        modules.append(ASPPPooling(in_channels, out_channels))
        self.convs = nn.ModuleList(modules)
        self.project = nn.Sequential(Conv2d(len(self.convs) * out_channels, out_channels, 1, bias=False),
                                     BatchNorm2d(out_channels), ReLU(), Dropout(0.5))

    def forward_nhwc(self, x):
        xs = ops.fan_out(x, len(self.convs))              # one-pass gradient fan-in over the five branches
        branches = [conv.forward_nhwc(xi) for conv, xi in zip(self.convs, xs)]
        y = ops.conv1x1_cat(self.project[0].weight, branches)
        y = self.project[1].forward_nhwc(y, relu=True)
        return self.project[3].forward_nhwc(y)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return ops.from_nhwc(self.forward_nhwc(ops.to_nhwc(x)))


class DeepLabHead(nn.Sequential):
    def __init__(self, in_channels: int, num_classes: int) -> None:
        super().__init__(ASPP(in_channels, [12, 24, 36]),
                         Conv2d(256, 256, 3, padding=1, bias=False),
                         BatchNorm2d(256),
                         ReLU(),
                         Conv2d(256, num_classes, 1))

    def forward_nhwc(self, x):
        y = self[0].forward_nhwc(x)
        y = conv_bn_act(y, self[1], self[2], relu=True)
        return self[4].forward_nhwc(y)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return ops.from_nhwc(self.forward_nhwc(ops.to_nhwc(x)))
