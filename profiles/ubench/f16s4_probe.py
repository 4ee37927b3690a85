#!/usr/bin/env python3
"""The two NT tile configurations side by side on the C2 step's short-reduction shapes: for each shape the 8-wave 256 x 128
kernel (GLF_F16S4 = 0 in a child process) against the 4-wave 128 x 128 / two-workgroups-per-CU kernel (GLF_F16S4 = 2), weights
pre-split, activations pre-split or not, plus a max-abs comparison of the two results.  Usage: f16s4_probe.py [reps]"""
import os
import subprocess
import sys
import time

HERE = os.path.abspath(__file__)
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(HERE))))
    import torch
    from glfusion_amd import ops
    reps = int(sys.argv[2])
    ops.set_precision("f16x3")
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(0)
    rnd = lambda *s: torch.rand(*s, device=dev, generator=g) * 2 - 1
    # (M, N, K, taps-as-3x3?)  plain 1x1 shapes of layers 1-3 and the heads, then 3x3 convs through conv2d
    PLAIN = [(193600, 256, 64), (193600, 64, 256), (193600, 64, 64), (50176, 512, 128), (50176, 128, 512), (50176, 1024, 256), (50176, 256, 1024),
             (50176, 2048, 512), (50176, 512, 2048), (50176, 256, 1280), (50176, 256, 2048), (150528, 3072, 2048)]
    for M, N, K in PLAIN:
        A, B, C = rnd(M, K), rnd(N, K), torch.empty(M, N, device=dev)
        ama, amb = ops.amax_of(A), ops.amax_of(B)
        Bp = ops.packed_of(B, amb)
        Ap = ops.packed_of(A, ama)
        for pa in (False, True):
            run = lambda: ops.gemm("nt", Ap if pa else A, Bp, C, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, amax_a=ama, amax_b=amb, a_packed=pa, b_packed=True)
            run(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                run()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / reps
            print(f"RES nt {M} {N} {K} pa={int(pa)} {dt * 1e3:.4f} {float(C.double().abs().sum()):.10e}", flush=True)
        del A, B, C, Ap, Bp
    for nimg, hh, cin, cout, dil in [(64, 55, 64, 64, 1), (64, 28, 128, 128, 1), (64, 28, 256, 256, 2)]:
        x = rnd(nimg, hh, hh, cin).requires_grad_(True)
        w = (rnd(cout, cin, 3, 3) / (3 * cin ** 0.5)).requires_grad_(True)
        gy = rnd(nimg, hh, hh, cout)
        fwd = lambda: ops.conv2d(x, w, None, 1, dil, dil)
        y = fwd(); y.backward(gy); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            y = fwd()
        torch.cuda.synchronize()
        tf = (time.perf_counter() - t0) / reps
        t0 = time.perf_counter()
        for _ in range(reps):
            y = fwd(); y.backward(gy); x.grad = None; w.grad = None
        torch.cuda.synchronize()
        tb = (time.perf_counter() - t0) / reps - tf
        print(f"RES conv3x3 {nimg * hh * hh} {cout} {cin} fwd {tf * 1e3:.4f} {float(y.double().abs().sum()):.10e}", flush=True)
        print(f"RES conv3x3 {nimg * hh * hh} {cout} {cin} bwd {tb * 1e3:.4f} 0", flush=True)
    sys.exit(0)

reps = sys.argv[1] if len(sys.argv) > 1 else "20"
res = {}
for mode in ("0", "2"):
    out = subprocess.run([sys.executable, HERE, "child", reps], env=dict(os.environ, GLF_F16S4=mode), capture_output=True, text=True, timeout=500)
    if out.returncode != 0:
        print(out.stderr[-2000:]); sys.exit(1)
    for line in out.stdout.splitlines():
        if line.startswith("RES "):
            p = line.split()
            res.setdefault(tuple(p[1:-2]), {})[mode] = (float(p[-2]), float(p[-1]))
print(f"{'shape':44s} {'8-wave ms':>10s} {'4-wave ms':>10s}  speed-up   |sum| rel diff")
for k, v in res.items():
    a, b = v["0"], v["2"]
    rel = abs(a[1] - b[1]) / max(abs(a[1]), 1e-30)
    print(f"{' '.join(k):44s} {a[0]:10.4f} {b[0]:10.4f}  {a[0] / b[0]:7.2f}x   {rel:.1e}")
