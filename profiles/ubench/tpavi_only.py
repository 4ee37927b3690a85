#!/usr/bin/env python3
"""The fusion block alone: one TPAVIModule(2048, mode='dot') at the C2 shape [64, 3, 28, 28, 2048], train-mode forward +
backward.  Prints ms per forward+backward and the FLOP-based rate; run under `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES
GRBM_GUI_ACTIVE` (their own pass) for the MFMA-busy fraction of the whole block.  Usage: tpavi_only.py [precision] [iters]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from glfusion_amd import ops
from glfusion_amd.models.ours import TPAVIModule

prec = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ops.set_precision(prec)
dev = torch.device("cuda", 0)
torch.manual_seed(0)
m = TPAVIModule(2048, mode="dot").to(dev).train()
torch.nn.init.normal_(m.W_z[1].weight, 1.0, 0.1)          # the reference zero-initialises this BatchNorm: make the block do work
n, v, h, w, c = 64, 3, 28, 28, 2048
x = (torch.randn(n, v, h, w, c, device=dev) * 0.5).requires_grad_(True)
gz = torch.randn(n, v, h, w, c, device=dev)


def run():
    for p in m.parameters():
        p.grad = None
    x.grad = None
    z = m.forward_nvhwc(x)
    z.backward(gz)


for _ in range(3):
    run()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(iters):
    run()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / iters
rows, L, ci = n * v * h * w, v * h * w, 1024
fwd = 2.0 * rows * c * 3 * ci + 2 * (2.0 * n * L * ci * ci) + 2.0 * rows * ci * c        # qkv, M = phi^T g, y = theta M, W_z
print(f"{prec}: {dt * 1e3:.2f} ms per forward+backward, {3 * fwd / dt / 1e12:.1f} TFLOP/s fp32-equivalent (executed, re-associated), "
      f"x3 products = {9 * fwd / dt / 1e12 / 2500:.3f} of the 2.5 PF fp16 MFMA peak", flush=True)
