#!/usr/bin/env python3
"""Which part of the step breaks hipGraph capture?  Runs each stage in a child process (a crash in hipStreamEndCapture kills
only the child).  Usage: graph_bisect.py            (driver)   |   graph_bisect.py STAGE   (one stage)"""
import faulthandler
import gc
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
STAGES = ["step_par01", "step_par012"]

if len(sys.argv) == 1:
    for st in STAGES:
        env = dict(os.environ, GLF_STREAMS="0" if st.endswith("_s0") else "1")
        t0 = time.time()
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), st], env=env, timeout=300, capture_output=True, text=True)
        except subprocess.TimeoutExpired:
            print(f"{st}: TIMEOUT -- stopping", flush=True)
            break
        tail = [l for l in (r.stdout + r.stderr).splitlines() if l.strip() and "amdgpu.ids" not in l][-6:]
        print(f"{st}: rc={r.returncode} ({time.time() - t0:.0f} s)\n    " + "\n    ".join(tail), flush=True)
    sys.exit(0)

faulthandler.enable()
sys.path.insert(0, ROOT)
import torch
import bench
from glfusion_amd import ops

stage = sys.argv[1]
ops.set_precision("f16x3")
dev = torch.device("cuda", 0)
cap = torch.cuda.Stream(device=dev)


def capture_and_replay(fn, reset):
    with torch.cuda.stream(cap):
        for _ in range(2):
            reset()
            fn()
        torch.cuda.synchronize()
        reset()
        gc.collect()
        ops._amax_pool.clear()
        ops._stats_pool.clear()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=cap):
            out = fn()
        print("captured", flush=True)
        g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            g.replay()
        torch.cuda.synchronize()
        print(f"replay ok: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms", flush=True)
    return out


if stage == "sections_small":
    from glfusion_amd.models.layers import Conv2d
    convs = [Conv2d(256, 256, 3, padding=1, bias=False).to(dev) for _ in range(3)]
    xs = [torch.randn(8, 28, 28, 256, device=dev, requires_grad=True) for _ in range(3)]

    def fn():
        ys = ops.parallel_sections([lambda c=c, x=x: c.forward_nhwc(x) for c, x in zip(convs, xs)])
        tot = ys[0] + ys[1] + ys[2]
        tot.backward(tot.detach())

    def reset():
        for c in convs:
            c.weight.grad = None
        for x in xs:
            x.grad = None
    capture_and_replay(fn, reset)
else:
    model = bench.build_model(dev)
    imgs, tgts = bench.make_batch(dev, 0, 64)

    def reset():
        for p in model.parameters():
            p.grad = None

    if stage.startswith("encoder"):
        def fn():
            f = model._encode(imgs)
            tot = None
            for v in bench.VIEWS:
                tot = f[v] if tot is None else tot + f[v]
            tot.backward(tot.detach())
    elif stage.startswith("head"):
        x = torch.randn(64, 28, 28, 2048, device=dev, requires_grad=True)

        def fn():
            x.grad = None
            y = model.classifier["1"].forward_nhwc(x)
            y.backward(y.detach())
    elif stage.startswith("tpavi"):
        x = torch.randn(64, 3, 28, 28, 2048, device=dev, requires_grad=True)

        def fn():
            x.grad = None
            y = model.global_attn.forward_nvhwc(x)
            y.backward(y.detach())
    elif stage.startswith("fwd_only"):
        def fn():
            with torch.no_grad():
                return model(imgs)[0]
    else:
        if "_par" in stage:                    # only the listed parallel_sections calls of a step (0 = view sections, 1 = fusion blocks, 2 = heads) fork
            allowed = {int(ch) for ch in stage.split("_par")[1]}
            real = ops.parallel_sections
            count = [0]

            def gated(fns):
                i = count[0]
                count[0] += 1
                return real(fns) if i in allowed else [f() for f in fns]
            import glfusion_amd.models.ours as ours_mod
            ops.parallel_sections = gated

        def fn():
            if "_par" in stage:
                count[0] = 0
            pred = model(imgs)[0]
            loss = None
            for v in bench.VIEWS:
                l = ops.bce_with_logits_sum(pred[v], tgts[v])
                loss = l if loss is None else loss + l
            loss.backward()
            return loss.detach()
    capture_and_replay(fn, reset)
