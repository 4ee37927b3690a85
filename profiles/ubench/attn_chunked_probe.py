#!/usr/bin/env python3
"""Softmax attention of TPAVIModule mode='embedded' as per-frame-group contractions on the split-fp16 kernels (fusion.py:
chunked_softmax_ok) -- the attention part alone, at the C2 shape (N = 64, L = 2352, Ci = 1024) and the config-5 length (L = 15 680,
N = 2): ms per pass and fp32-equivalent TFLOP/s (forward 2 x 2 N L^2 Ci; backward 5 x: S recomputed, dP, dg, dtheta, dphi).
Usage: attn_chunked_probe.py [iters]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from glfusion_amd import fusion, ops
from glfusion_amd._lib import check, lib

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = torch.device("cuda", 0)
ops.set_precision("f16x3")
gen = torch.Generator(device=dev).manual_seed(0)
p = ops._p
for n, L, ci in ((64, 2352, 1024), (2, 15680, 1024)):
    rows, c3 = n * L, 3 * ci
    lp = (L + 31) // 32 * 32
    qkv = (torch.rand(rows, c3, device=dev, generator=gen) - 0.5) * 0.2
    th, ph, g = qkv[:, :ci], qkv[:, ci:2 * ci], qkv[:, 2 * ci:]
    am_q = ops.amax_of(qkv)
    y = torch.empty(rows, ci, device=dev)
    dy = torch.rand(rows, ci, device=dev, generator=gen) - 0.5
    am_dy = ops.amax_of(dy)
    dqkv = torch.empty_like(qkv)
    dth, dph, dg = dqkv[:, :ci], dqkv[:, ci:2 * ci], dqkv[:, 2 * ci:]
    gpc = fusion._frames_per_chunk(n, L)
    S = torch.empty(gpc, L, lp, device=dev)
    dP = torch.empty(gpc, L, lp, device=dev)
    T = torch.empty(gpc, ci, lp, device=dev)
    one = ops._ones4(dev)[:1]
    bq, bs = L * c3, L * ci

    def fwd():
        for f0 in range(0, n, gpc):
            gc = min(gpc, n - f0)
            fusion._scores(th, ph, f0, gc, L, lp, ci, c3, am_q, S)
            fusion._transposed(g, f0, gc, L, lp, ci, c3, T)
            ops.gemm("nt", S, T, y[f0 * L:], M=L, N=ci, K=lp, lda=lp, ldb=lp, ldc=ci, batch=gc, bsa=L * lp, bsb=ci * lp, bsc=L * ci, amax_a=one, amax_b=am_q)

    def bwd():
        for f0 in range(0, n, gpc):
            gc = min(gpc, n - f0)
            fusion._scores(th, ph, f0, gc, L, lp, ci, c3, am_q, S)
            am_dP = ops.amax_slot(dev)
            ops.gemm("nt", dy[f0 * L:], g[f0 * L:], dP, M=L, N=L, K=ci, lda=ci, ldb=c3, ldc=lp, batch=gc, bsa=bs, bsb=bq, bsc=L * lp, amax_a=am_dy, amax_b=am_q, amax_c=am_dP)
            ops.gemm("tn", S, dy[f0 * L:], dg[f0 * L:], M=L, N=ci, K=L, lda=lp, ldb=ci, ldc=c3, batch=gc, bsa=L * lp, bsb=bs, bsc=bq, amax_a=one, amax_b=am_dy)
            check(lib.glf_softmax_rows_bwd_ld(p(S), p(dP), gc * L, L, lp, ops._stream()), "sb")
            am_dS = ops.amax_slot(dev)
            check(lib.glf_amax_combine(p(am_dP), None, 2.0, 0, p(am_dS), ops._stream()), "comb")
            fusion._transposed(ph, f0, gc, L, lp, ci, c3, T)
            ops.gemm("nt", dP, T, dth[f0 * L:], M=L, N=ci, K=lp, lda=lp, ldb=lp, ldc=c3, batch=gc, bsa=L * lp, bsb=ci * lp, bsc=bq, amax_a=am_dS, amax_b=am_q)
            ops.gemm("tn", dP, th[f0 * L:], dph[f0 * L:], M=L, N=ci, K=L, lda=lp, ldb=c3, ldc=c3, batch=gc, bsa=L * lp, bsb=bq, bsc=bq, amax_a=am_dS, amax_b=am_q)

    print(f"frames per group {gpc} ({gpc * L * lp * 4 / 2**30:.2f} GiB of scores alive, x2 in backward)")
    for fn, name, mults in ((fwd, "forward ", 2), (bwd, "backward", 5)):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / iters
        fl = mults * 2.0 * n * L * L * ci
        print(f"N={n} L={L} Ci={ci} {name}: {dt * 1e3:8.2f} ms  {fl / dt / 1e12:6.1f} TFLOP/s fp32-equivalent ({3 * fl / dt / 1e12:6.1f} executed fp16 MFMA)", flush=True)
    del qkv, y, dy, dqkv, S, dP, T
