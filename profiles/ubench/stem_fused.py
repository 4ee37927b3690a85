#!/usr/bin/env python3
"""Inference-mode init_block at the bench's frame count (192 frames of 112 x 112): glf_stem7x7_bn_relu_pool against the conv /
BatchNorm / max-pool chain it replaces under no_grad (HIP events, 20 launches each after 3)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from glfusion_amd import ops
from glfusion_amd.models.layers import BatchNorm2d, Conv2d, MaxPool2d, init_block_nhwc

dev = "cuda"
conv, bn, pool = Conv2d(1, 64, kernel_size=7, stride=1, padding=2).to(dev).eval(), BatchNorm2d(64).to(dev).eval(), MaxPool2d(3, 2, 1)
for n, hw in ((192, 112), (160, 224)):
    x = ops.to_nhwc(torch.rand(n, 1, hw, hw, device=dev))
    for fused in (False, True, False, True):
        ops.FUSED_STEM = fused
        with torch.no_grad():
            for _ in range(3):
                y = init_block_nhwc(x, conv, bn, pool)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                y = init_block_nhwc(x, conv, bn, pool)
            e1.record()
        e1.synchronize()
        ho = hw - 2
        moved = n * (hw * hw + (0 if fused else 2 * 2 * ho * ho * 64) + ((ho - 1) // 2 + 1) ** 2 * 64 + (0 if fused else ho * ho * 64)) * 4
        print(f"{n} x {hw}^2  {'one launch   ' if fused else 'three kernels'}  {e0.elapsed_time(e1) / 20:.3f} ms   tensor bytes moved {moved / 1e9:.2f} GB")
