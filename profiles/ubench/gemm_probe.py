#!/usr/bin/env python3
"""Runs a few of the C2 step's contraction shapes in isolation (random operands) -- the target of rocprofv3 counter
passes and of A/B timing between library builds (GLF_LIB_PATH).  Usage: gemm_probe.py [precision] [reps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from glfusion_amd import ops

prec = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
ops.set_precision(prec)
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s: torch.rand(*s, device=dev, generator=g) * 2 - 1

SHAPES = [("nt", 150528, 3072, 2048), ("nt", 150528, 2048, 1024), ("nt", 50176, 2048, 512), ("nt", 50176, 512, 2048),
          ("nt", 50176, 256, 1024), ("nt", 50176, 1024, 256), ("tn", 3072, 2048, 150528), ("tn", 2048, 512, 50176)]
CONVS = [(64, 28, 256, 256, 1), (64, 28, 512, 512, 2), (64, 28, 2048, 256, 12), (64, 28, 256, 256, 2), (64, 55, 64, 64, 1), (64, 28, 128, 128, 1)]
if os.environ.get("PROBE_CONV"):                  # 3x3 convs through the autograd op instead: forward, dgrad + wgrad (ms each side)
    import torch.nn.functional as F
    for nimg, hh, cin, cout, dil in CONVS:
        x = rnd(nimg, hh, hh, cin).requires_grad_(True)
        w = (rnd(cout, cin, 3, 3) / (3 * cin ** 0.5)).requires_grad_(True)
        gy = rnd(nimg, hh, hh, cout)
        def fwd():
            return ops.conv2d(x, w, None, 1, dil, dil)
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
        fl = 2.0 * nimg * hh * hh * cin * cout * 9
        print(f"{prec} conv3x3 {nimg}x{hh}x{hh} {cin}->{cout} dil {dil}: fwd {tf * 1e3:.3f} ms ({fl / tf / 1e12:.0f} TF dense), bwd {tb * 1e3:.3f} ms ({2 * fl / tb / 1e12:.0f} TF dense)", flush=True)
    sys.exit(0)
PACK = os.environ.get("PROBE_PACK", "")            # e.g. "ab": which operands go in pre-split
LDPAD = int(os.environ.get("PROBE_LDPAD", "0"))    # NT shapes: row stride K + LDPAD floats (is a power-of-two stride a problem?)
if os.environ.get("PROBE_SHAPES"):
    SHAPES = [SHAPES[int(i)] for i in os.environ["PROBE_SHAPES"].split(",")]
for mode, M, N, K in SHAPES:
    if mode == "nt":
        A, B, C = rnd(M, K + LDPAD), rnd(N, K + LDPAD), torch.empty(M, N, device=dev)
    else:
        A, B, C = rnd(K, M), rnd(K, N), torch.zeros(M, N, device=dev)
    ama, amb = ops.amax_of(A), ops.amax_of(B)
    pa, pb = "a" in PACK and ama is not None, "b" in PACK and amb is not None
    Ax = ops.packed_of(A, ama) if pa else A
    Bx = ops.packed_of(B, amb) if pb else B
    if mode == "nt":
        run = lambda X=Ax, Y=Bx, pa=pa, pb=pb: ops.gemm("nt", X, Y, C, M=M, N=N, K=K, lda=K + LDPAD, ldb=K + LDPAD, ldc=N, amax_a=ama, amax_b=amb, a_packed=pa, b_packed=pb)
    else:
        sp = ops._tn_split(K, M, N, 1)
        run = lambda X=Ax, Y=Bx, pa=pa, pb=pb: ops.gemm("tn", X, Y, C, M=M, N=N, K=K, lda=M, ldb=N, ldc=N, split=sp, amax_a=ama, amax_b=amb, a_packed=pa, b_packed=pb)
    if (pa or pb) and not os.environ.get("PROBE_NOCHECK"):   # the pre-split path must reproduce the in-kernel split bit for bit
        run(A, B, False, False)
        ref = C.clone()
        run()
        assert torch.equal(ref, C), f"pre-split result differs: {(ref - C).abs().max().item()}"
    run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"{prec} pack={PACK or '-'} ldpad={LDPAD} {mode} M={M} N={N} K={K}: {dt * 1e3:.3f} ms  {2.0 * M * N * K / dt / 1e12:.1f} TF fp32-equivalent", flush=True)
    del A, B, C, Ax, Bx
