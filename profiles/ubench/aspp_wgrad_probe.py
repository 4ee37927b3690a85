#!/usr/bin/env python3
"""The ASPP 3x3 dilated convs' weight gradient alone (2048 -> 256 on 64 x 28 x 28, rates 12 / 24): per-tap rectangles with
different slice counts against the dense form.  Usage: aspp_wgrad_probe.py [reps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from glfusion_amd import ops

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
ops.set_precision("f16x3")
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s: torch.rand(*s, device=dev, generator=g) * 2 - 1
nimg, hh, cin, cout = 64, 28, 2048, 256
x = rnd(nimg, hh, hh, cin)                     # no gradient: backward = the weight gradient only
gy = rnd(nimg, hh, hh, cout)
real_split = ops.wgrad_split
for dil in (12, 24):
    w = (rnd(cout, cin, 3, 3) / (3 * cin ** 0.5)).requires_grad_(True)
    for label, thr, scale in (("rect, default slices", 0.8, 1.0), ("rect, slices x0.5", 0.8, 0.5), ("rect, slices x2", 0.8, 2.0), ("rect, slices x4", 0.8, 4.0),
                              ("dense (all taps, full rows)", 0.0, 1.0)):
        ops.RECT_THRESHOLD["wgrad"] = thr
        ops.wgrad_split = lambda rows_o, frac, co, ci, ntap, rect, s=scale: max(1, min(int(real_split(rows_o, frac, co, ci, ntap, rect) * s), rows_o // 512))
        def run():
            w.grad = None
            y = ops.conv2d(x, w, None, 1, dil, dil)
            y.backward(gy)
        def fwd_only():
            with torch.no_grad():
                ops.conv2d(x, w, None, 1, dil, dil)
        for fn in (run, fwd_only):
            fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            run()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(reps):
            fwd_only()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        wg = ((t1 - t0) - (t2 - t1)) / reps
        print(f"dil {dil:2d} {label:30s}: wgrad {wg * 1e3:7.3f} ms", flush=True)
ops.wgrad_split = real_split
