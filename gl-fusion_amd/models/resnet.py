"""ResNet-50 trunk the reference takes from torchvision==0.9.1 (requirements.txt:25; call site
models/segmentation.py:205-207 with replace_stride_with_dilation=[False, True, True]).
torchvision is not vendored in the reference tree, so this is a restatement of the published
v1.5 architecture with the same attribute names (=> the same state_dict keys); the in-tree
structural pin is models/resnet.py:43-79."""
from __future__ import annotations

from typing import Optional, Sequence

import torch
from torch import nn

from .. import ops
from .layers import BatchNorm2d, Conv2d, MaxPool2d, ReLU, conv_bn_act


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes: int, planes: int, stride: int = 1, downsample: Optional[nn.Module] = None,
                 dilation: int = 1) -> None:
        super().__init__()
        self.conv1 = Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = BatchNorm2d(planes)
        self.conv2 = Conv2d(planes, planes, 3, stride=stride, padding=dilation, dilation=dilation, bias=False)
        self.bn2 = BatchNorm2d(planes)
        self.conv3 = Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = BatchNorm2d(planes * 4)
        self.relu = ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward_nhwc(self, x, sole_reader: bool = False):
        # x feeds the shortcut and conv1: its two gradients meet in ONE pass (ops.fan_out) instead of autograd's add -- or, when
        # x is the previous block's output and nobody else reads it (sole_reader, set by Stage), in no pass of their own at all:
        # that block's last BatchNorm backward adds them while reading
        x, xs = ops.fan_out(x, 2, lazy=sole_reader)
        if self.downsample is None and x is not xs:
            ops.join_gradients(x, xs)     # identity shortcut: conv1's dgrad accumulates onto the shortcut's gradient
        idn = xs if self.downsample is None else conv_bn_act(xs, self.downsample[0], self.downsample[1], relu=False)
        out = conv_bn_act(x, self.conv1, self.bn1, relu=True, consumer=self.conv2)      # inner activations: one reader each
        out = conv_bn_act(out, self.conv2, self.bn2, relu=True, consumer=self.conv3)
        return conv_bn_act(out, self.conv3, self.bn3, relu=True, residual=idn)      # relu(bn3(.) + identity)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return ops.from_nhwc(self.forward_nhwc(ops.to_nhwc(x)))


class Stage(nn.Sequential):
    """layerN: a Sequential of Bottlenecks that stays NHWC between its blocks."""

    def forward_nhwc(self, x, sole_reader: bool = False):
        """sole_reader: x is the output of the previous stage's last block (a BatchNorm + residual + ReLU) and this stage is its
        only reader -- never true for layer1, whose input comes out of the max-pool."""
        for i, blk in enumerate(self):
            x = blk.forward_nhwc(x, sole_reader=sole_reader or i > 0)       # block i > 0 reads block i - 1's output, which nobody else sees
        return x

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return ops.from_nhwc(self.forward_nhwc(ops.to_nhwc(x)))


class ResNet(nn.Module):
    def __init__(self, layers: Sequence[int] = (3, 4, 6, 3), replace_stride_with_dilation: Sequence[bool] = (False, False, False),
                 num_classes: int = 1000) -> None:
        super().__init__()
        self.inplanes, self.dilation = 64, 1
        self.conv1 = Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)   # replaced by _utils.py:192
        self.bn1 = BatchNorm2d(64)
        self.relu = ReLU(inplace=True)
        self.maxpool = MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._make_layer(64, layers[0], 1, False)
        self.layer2 = self._make_layer(128, layers[1], 2, replace_stride_with_dilation[0])
        self.layer3 = self._make_layer(256, layers[2], 2, replace_stride_with_dilation[1])
        self.layer4 = self._make_layer(512, layers[3], 2, replace_stride_with_dilation[2])
        self.avgpool = nn.AdaptiveAvgPool2d(1)          # parameter-free; dropped by IntermediateLayerGetter
        self.fc = nn.Linear(2048, num_classes)          # idem (never reached on the path)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def _make_layer(self, planes: int, blocks: int, stride: int, dilate: bool) -> Stage:
        prev = self.dilation
        if dilate:
            self.dilation *= stride
            stride = 1
        downsample = None
        if stride != 1 or self.inplanes != planes * 4:
            downsample = nn.Sequential(Conv2d(self.inplanes, planes * 4, 1, stride=stride, bias=False), BatchNorm2d(planes * 4))
        blks = [Bottleneck(self.inplanes, planes, stride, downsample, prev)]
        self.inplanes = planes * 4
        blks += [Bottleneck(self.inplanes, planes, dilation=self.dilation) for _ in range(1, blocks)]
        return Stage(*blks)


def resnet50(pretrained: bool = False, progress: bool = True, **kwargs) -> ResNet:
    if pretrained:
        # the reference passes pretrained_backbone=True by default (segmentation.py:202) and would download
        # ImageNet weights; there is no network on the target, checkpoints are loaded via load_state_dict
        pass
    return ResNet((3, 4, 6, 3), kwargs.get("replace_stride_with_dilation", (False, False, False)))
