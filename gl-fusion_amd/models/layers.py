"""nn.Module leaves with the reference's parameter names and default initialisation, whose
forward runs on the HIP engine.  They subclass the torch modules only to inherit parameter
registration / state_dict behaviour; no torch compute kernel is used in forward.

Convention between these modules: tensors are NCHW-shaped with channels_last strides
(ops.from_nhwc), so they look like ordinary torch tensors to callers while the kernels see
row-major [N*H*W, C] matrices.
"""
from __future__ import annotations

import os

import torch
from torch import nn

from .. import ops


def _one(v) -> int:
    if isinstance(v, (tuple, list)):
        if any(int(e) != int(v[0]) for e in v):
            raise RuntimeError(f"glfusion_amd: anisotropic conv/pool parameter {v} is not on the path")
        return int(v[0])
    return int(v)


class Conv2d(nn.Conv2d):
    def _geom(self):
        if self.groups != 1 or self.padding_mode != "zeros" or isinstance(self.padding, str):
            raise RuntimeError("glfusion_amd: only groups=1, zero padding convolutions are on the path")
        return _one(self.stride), _one(self.padding), _one(self.dilation)

    def forward_nhwc(self, x: torch.Tensor, colstats=None) -> torch.Tensor:
        stride, pad, dil = self._geom()
        if self.in_channels == 1 and self.kernel_size == (7, 7) and stride == 1 and dil == 1:
            return ops.stem7x7(x, self.weight, self.bias, pad)          # models/_utils.py:192
        if self.in_channels % 4 != 0:
            raise RuntimeError(f"glfusion_amd: conv with Cin={self.in_channels} is not on the path")
        return ops.conv2d(x, self.weight, self.bias, stride, pad, dil, colstats)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return ops.from_nhwc(self.forward_nhwc(ops.to_nhwc(x)))


class BatchNorm2d(nn.BatchNorm2d):
    def forward_nhwc(self, x, relu: bool = False, residual=None, sums=None, packed_grad: bool = False, packed_out: bool = False):
        return ops.batch_norm_act(x, self, relu, residual, sums, packed_grad, packed_out)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return ops.from_nhwc(self.forward_nhwc(ops.to_nhwc(x)))


class ReLU(nn.ReLU):
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return ops.from_nhwc(ops.relu(ops.to_nhwc(x)))


class MaxPool2d(nn.MaxPool2d):
    def forward_nhwc(self, x):
        if (_one(self.kernel_size), _one(self.stride), _one(self.padding), _one(self.dilation)) != (3, 2, 1, 1) or self.ceil_mode:
            raise RuntimeError("glfusion_amd: only MaxPool2d(3, stride=2, padding=1) is on the path")
        return ops.maxpool3x3s2(x)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return ops.from_nhwc(self.forward_nhwc(ops.to_nhwc(x)))


class Dropout(nn.Dropout):
    def forward_nhwc(self, x):
        return ops.dropout(x, self.p, self.training)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return ops.from_nhwc(self.forward_nhwc(ops.to_nhwc(x)))


class AdaptiveAvgPool2d(nn.AdaptiveAvgPool2d):
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if _one(self.output_size) != 1:
            raise RuntimeError("glfusion_amd: only AdaptiveAvgPool2d(1) is on the path")
        return ops.from_nhwc(ops.global_avgpool(ops.to_nhwc(x)))


FUSE_BN_STATS = os.environ.get("GLF_FUSE_BN_STATS", "1") != "0"


def conv_bn_act(x, conv: Conv2d, bn: BatchNorm2d, relu: bool, residual=None, consumer: Conv2d = None):
    """conv -> BatchNorm (train or eval) -> (+residual) -> (ReLU) on NHWC tensors.  In train() the batch statistics
    (sum x, sum x^2 per channel) are accumulated by the conv's own epilogue where the kernel supports it, which saves the
    separate statistics pass over the conv output.
    consumer: the convolution that is the ONLY reader of the result (the next conv inside a bottleneck).  Where its kernels
    take a packed pre-split input, the result is written once in that form and has no fp32 copy: do not read it as floats."""
    training = bn.training or bn.running_mean is None
    stem = conv.in_channels == 1 and conv.kernel_size == (7, 7)
    # the conv output's gradient has ONE consumer, this conv's backward: BatchNorm backward may hand it over as a packed image
    pg = (not stem) and conv.bias is None and torch.is_grad_enabled() and ops.takes_packed_grad(conv.weight)
    if training and FUSE_BN_STATS and not stem:
        stride, pad, dil = conv._geom()
        if ops.conv_stats_fusable(conv.weight, stride, pad, dil, x.shape[1], x.shape[2], x.dtype):
            sums = ops.stats_slot(conv.out_channels, x.device)
            po = consumer is not None and residual is None and ops.takes_packed_input(consumer.weight)
            if po:
                sums._glf_colmax = ops.colmax_slot(conv.out_channels, x.device)     # bounds the BatchNorm output before it exists
            return bn.forward_nhwc(conv.forward_nhwc(x, sums), relu=relu, residual=residual, sums=sums, packed_grad=pg, packed_out=po)
    return bn.forward_nhwc(conv.forward_nhwc(x), relu=relu, residual=residual, packed_grad=pg)


def init_block_nhwc(x, conv: Conv2d, bn: BatchNorm2d, pool: MaxPool2d):
    """init_block = conv1 (7x7, Cin 1), bn1, relu, maxpool (ours.py:1725-1730, 1796) on an NHWC [N,H,W,1] input.  Evaluation under
    no_grad with running statistics: ONE launch (glf_stem7x7_bn_relu_pool; the conv output never reaches memory).  Training, or any
    call that records a graph: conv -> BatchNorm (batch statistics) -> ReLU -> max-pool as three kernels with their backward."""
    stride, pad, dil = conv._geom()
    fused = (ops.FUSED_STEM and not ops.s16() and not torch.is_grad_enabled() and not (bn.training or bn.running_mean is None)
             and conv.in_channels == 1 and conv.out_channels == 64 and conv.kernel_size == (7, 7) and stride == 1 and dil == 1 and pad <= 3
             and (_one(pool.kernel_size), _one(pool.stride), _one(pool.padding), _one(pool.dilation)) == (3, 2, 1, 1) and not pool.ceil_mode)
    if fused:
        return ops.stem_bn_relu_pool(x, conv.weight, conv.bias, bn.running_mean, bn.running_var, bn.weight, bn.bias, float(bn.eps), pad)
    return pool.forward_nhwc(conv_bn_act(x, conv, bn, relu=True))
