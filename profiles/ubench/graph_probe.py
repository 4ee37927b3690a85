#!/usr/bin/env python3
"""Feasibility probe: capture the C2 train step (forward + sum-BCE + backward) in ONE hipGraph through torch.cuda.CUDAGraph
and compare replay time with the eager step.  Usage: graph_probe.py [precision] [steps]"""
import faulthandler
import gc
import os
import sys
import time

faulthandler.enable()

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from glfusion_amd import ops

prec = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
ops.set_precision(prec)
dev = torch.device("cuda", 0)
model = bench.build_model(dev)
imgs, tgts = bench.make_batch(dev, 0, 64)


def step():
    for p in model.parameters():
        p.grad = None
    pred = model(imgs)[0]
    loss = None
    for v in bench.VIEWS:
        l = ops.bce_with_logits_sum(pred[v], tgts[v])
        loss = l if loss is None else loss + l
    loss.backward()
    return loss


def timed(fn, k):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e3, out


cap = torch.cuda.Stream(device=dev)            # every lazily-created per-stream resource (library rings, pools) is warmed up on it
cap.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(cap):
    for _ in range(3):
        step()
    ms, loss = timed(step, steps)
    print(f"eager   : {ms:.1f} ms/step, loss {float(loss.detach()):.4f}, streams={ops.STREAMS}", flush=True)
    gref = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    del loss
    for p in model.parameters():
        p.grad = None
torch.cuda.synchronize()
gc.collect()

# stage 1: a trivial captured launch through the library; stage 2: one conv forward + backward; stage 3: the step
with torch.cuda.stream(cap):
    g0 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g0, stream=cap):
        z = ops.zeros(1024, device=dev)
    g0.replay(); torch.cuda.synchronize()
    print("stage 1 (glf_zero captured) ok", flush=True)
    from glfusion_amd.models.layers import Conv2d
    conv = Conv2d(256, 256, 3, padding=1, bias=False).to(dev)
    xin = torch.randn(8, 28, 28, 256, device=dev, requires_grad=True)
    def cstep():
        conv.weight.grad = None; xin.grad = None
        y = conv.forward_nhwc(xin)
        y.backward(y.detach())
    cstep(); cstep(); torch.cuda.synchronize()
    conv.weight.grad = None; xin.grad = None
    ops._amax_pool.clear(); ops._stats_pool.clear()
    g1 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g1, stream=cap):
        cstep()
    g1.replay(); torch.cuda.synchronize()
    print("stage 2 (conv fwd+bwd captured) ok", flush=True)

# pools whose zero fill must be part of the captured work
ops._amax_pool.clear()
ops._stats_pool.clear()
g = torch.cuda.CUDAGraph()
t0 = time.perf_counter()
with torch.cuda.stream(cap):
    with torch.cuda.graph(g, stream=cap):
        gl = step()
torch.cuda.synchronize()
print(f"capture : {(time.perf_counter() - t0) * 1e3:.0f} ms host, reserved {torch.cuda.memory_reserved() / 2**30:.1f} GB", flush=True)
g.replay()
torch.cuda.synchronize()
print(f"replay 1: loss {float(gl):.4f}", flush=True)
ms, _ = timed(g.replay, steps)
print(f"graph   : {ms:.1f} ms/step, loss {float(gl):.4f}", flush=True)
worst = 0.0
for n, p in model.named_parameters():
    if n in gref:
        d = float((p.grad - gref[n]).norm() / (gref[n].norm() + 1e-30))
        worst = max(worst, d)
print(f"max rel-L2 grad deviation graph vs eager (different dropout masks expected): {worst:.3e}", flush=True)
