#!/usr/bin/env python3
"""BASELINE.json configs[4] on ONE GPU: 5 views x T frames x 224 x 224 (h = w = 56, L = V h w = 15 680 positions per frame),
train() forward + sum-BCE + backward.  Usage: config5.py [T=32] [precision=f16x3]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from glfusion_amd import ops
from glfusion_amd.models import Global_and_Local

T = int(sys.argv[1]) if len(sys.argv) > 1 else 32
prec = sys.argv[2] if len(sys.argv) > 2 else "f16x3"
ops.set_precision(prec)
dev = torch.device("cuda", 0)
views, H = ["1", "2", "3", "4", "5"], 224
torch.manual_seed(0)
model = Global_and_Local(views)
with torch.no_grad():
    for a in (model.global_attn, model.local_attn):
        a.W_z[1].weight.normal_(1.0, 0.1)
model = model.to(dev).train()
g = torch.Generator(device=dev).manual_seed(1)
imgs = {v: torch.rand(T, 1, H, H, device=dev, generator=g) for v in views}
tg = {v: (torch.rand(T, 5, H, H, device=dev, generator=g) < 0.3).float() for v in views}


def step():
    for p in model.parameters():
        p.grad = None
    pred = model(imgs)[0]
    loss = None
    for v in views:
        l = ops.bce_with_logits_sum(pred[v], tg[v])
        loss = l if loss is None else loss + l
    loss.backward()
    return loss


gb = 2 ** 30
for i in range(3):                                  # untimed: the first steps size the allocator's pools (and decide whether pre-split
    ts = time.perf_counter()                        # images can be retained for the backward pass, ops.retain_ok)
    step()
    torch.cuda.synchronize()
    print(f"  warm-up step {i}: {(time.perf_counter() - ts) * 1e3:.0f} ms, allocated peak {torch.cuda.max_memory_allocated() / gb:.1f} GB, "
          f"reserved {torch.cuda.memory_reserved() / gb:.1f} GB, retention off: {bool(ops._retain_off)}", flush=True)
t0 = time.perf_counter()
for _ in range(2):
    l = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 2
print(f"config-5 shape [{prec}]: 5 views x {T} x 224^2, L = {5 * 56 * 56}: {dt * 1e3:.0f} ms/step = {T / 32 / dt:.3f} clips/s (32-frame clips), "
      f"loss {float(l):.1f}, peak memory {torch.cuda.max_memory_allocated() / 2 ** 30:.1f} GB; dense 504.6 TFLOP/clip -> "
      f"{504.6 * T / 32 / dt:.0f} TFLOP/s dense-equivalent", flush=True)
