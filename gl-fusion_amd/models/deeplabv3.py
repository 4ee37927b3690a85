"""DeepLabV3 decoder of the reference (GLfusion/models/deeplabv3.py:102-166) on the HIP engine.

ASPP specifics of this build: the five branch outputs are never concatenated -- the 1280->256
projection is evaluated as five accumulated K=256 contractions (ops.conv1x1_cat); dilated
taps that fall into the zero padding for a whole tile are skipped inside the conv kernel
(rate 36 on a 28x28 map reduces to its centre tap); the pooled branch is a per-frame row
broadcast."""
from __future__ import annotations

import os
from typing import List

import torch
from torch import nn

from .. import ops
from .layers import AdaptiveAvgPool2d, BatchNorm2d, Conv2d, Dropout, ReLU, conv_bn_act
from ._utils import _SimpleSegmentationModel_iekd

_CAT_BUFFER = os.environ.get("GLF_ASPP_CAT", "1") != "0"

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

    def trunk_nhwc(self, x):
        """Everything up to (not including) the Dropout: branches, projection, BN, ReLU."""
        xs = ops.fan_out(x, len(self.convs))              # one-pass gradient fan-in over the five branches
        # every branch's last kernel (BatchNorm apply / broadcast) writes its 256 columns of ONE [N,h,w,1280] buffer, so
        # the 1280 -> 256 projection is a single K = 1280 contraction (and one dgrad, one wgrad) -- still no concat copy
        n, h, w = x.shape[0], x.shape[1], x.shape[2]
        widths = [conv[0].out_channels if not isinstance(conv, ASPPPooling) else conv[1].out_channels for conv in self.convs]
        cat = torch.empty(n, h, w, sum(widths), dtype=x.dtype, device=x.device)
        slot = ops.amax_slot(x.device)
        branches, off = [], 0
        for conv, xi, ck in zip(self.convs, xs, widths):
            if _CAT_BUFFER:
                with ops.output_into(cat[..., off:off + ck], slot):
                    branches.append(conv.forward_nhwc(xi))
            else:
                branches.append(conv.forward_nhwc(xi))
            off += ck
        pbn = self.project[1]
        if ops.s16() and _CAT_BUFFER and (pbn.training or pbn.running_mean is None):
            # 16-bit storage: the projection's BatchNorm statistics come out of its own epilogue
            sums = ops.stats_slot(self.project[0].out_channels, x.device)
            y = ops.conv1x1_cat(self.project[0].weight, branches, colstats=sums)
            return pbn.forward_nhwc(y, relu=True, sums=sums)
        y = ops.conv1x1_cat(self.project[0].weight, branches)
        # (the projection's gradient is consumed by ConvCatFn's dgrad / wgrad only -- and only its single-buffer form reads it packed)
        pg = _CAT_BUFFER and torch.is_grad_enabled() and ops.takes_packed_grad(self.project[0].weight)
        return self.project[1].forward_nhwc(y, relu=True, packed_grad=pg)

    def forward_nhwc(self, x):
        return self.project[3].forward_nhwc(self.trunk_nhwc(x))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return ops.from_nhwc(self.forward_nhwc(ops.to_nhwc(x)))


class DeepLabHead(nn.Sequential):
    def __init__(self, in_channels: int, num_classes: int) -> None:
        super().__init__(ASPP(in_channels, [12, 24, 36]),
                         Conv2d(256, 256, 3, padding=1, bias=False),
                         BatchNorm2d(256),
                         ReLU(),
                         Conv2d(256, num_classes, 1))

    def _tail_nhwc(self, trunk):
        y = self[0].project[3].forward_nhwc(trunk)        # Dropout: a fresh mask per call
        y = conv_bn_act(y, self[1], self[2], relu=True)
        return self[4].forward_nhwc(y)

    def forward_nhwc(self, x):
        return self._tail_nhwc(self[0].trunk_nhwc(x))

    def forward_nhwc_shared(self, x):
        """forward_nhwc(x) plus a token with which a LATER forward over the SAME input can be evaluated without
        recomputing the ASPP trunk (Global_and_Local applies a view's classifier to f4 twice, ours.py:1806 and
        1840): everything before the Dropout is a deterministic function of (input, weights) -- in train() the
        BatchNorm layers normalise with the batch statistics of that same input -- so the second call shares the
        trunk activation (its gradient becomes the sum of both uses) and differs only in what comes after:
        its own Dropout mask, tail conv / BN / output conv, and one more running-statistics update of the trunk's
        BatchNorm layers, which is replayed from the recorded batch statistics at the time of the second call."""
        prev, ops.BN_TAP = ops.BN_TAP, []
        try:
            trunk = self[0].trunk_nhwc(x)
            records = ops.BN_TAP
        finally:
            ops.BN_TAP = prev
        t1, t2 = ops.fan_out(trunk, 2)
        out = self._tail_nhwc(t1)
        return out, (t2, records, out)

    def forward_nhwc_replay(self, token):
        trunk, records, first_out = token
        if not self.training:
            return first_out                              # eval(): no Dropout, frozen statistics -- the same tensor
        ops.replay_bn_updates(records)
        return self._tail_nhwc(trunk)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return ops.from_nhwc(self.forward_nhwc(ops.to_nhwc(x)))
