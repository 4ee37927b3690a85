#!/usr/bin/env python3
"""The fused softmax attention kernels alone (TPAVIModule mode='embedded': glf_attn_softmax_fwd / _bwd) at the C2 shape
(N = 64 frames, L = 2352, Ci = 1024) and the config-5 shape (L = 15 680, N = 2): ms per pass, TFLOP/s and fraction of the exact-fp32
MFMA peak (157.3 TF) -- the kernel runs v_mfma_f32_32x32x2_f32.  Run under rocprofv3 --kernel-trace --stats / --pmc
SQ_VALU_MFMA_BUSY_CYCLES for the per-kernel rows.  Usage: attn_softmax_only.py [iters]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from glfusion_amd._lib import AttnParams, check, lib

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
for n, L, ci in ((64, 2352, 1024), (2, 15680, 1024)):
    rows = n * L
    qkv = (torch.rand(rows, 3 * ci, device=dev, generator=g) - 0.5) * 0.2
    th, ph, gg = qkv[:, :ci], qkv[:, ci:2 * ci], qkv[:, 2 * ci:]
    y = torch.empty(rows, ci, device=dev)
    lse = torch.empty(rows, device=dev)
    dy = torch.rand(rows, ci, device=dev, generator=g) - 0.5
    dqkv = torch.empty_like(qkv)
    dsum = torch.empty(rows, device=dev)
    ap = AttnParams()
    ap.frames, ap.L, ap.ci = n, L, ci
    ap.ldq = ap.ldk = ap.ldv = 3 * ci
    ap.ldy, ap.lddy, ap.ldd = ci, ci, 3 * ci
    p = lambda t: C.c_void_p(t.data_ptr())
    fwd = lambda: check(lib.glf_attn_softmax_fwd(p(th), p(ph), p(gg), p(y), p(lse), C.byref(ap), None), "fwd")
    bwd = lambda: check(lib.glf_attn_softmax_bwd(p(th), p(ph), p(gg), p(y), p(dy), p(lse), p(dqkv[:, :ci]), p(dqkv[:, ci:2 * ci]), p(dqkv[:, 2 * ci:]),
                                                 p(dsum), C.byref(ap), None), "bwd")
    for fn, name, mults in ((fwd, "forward ", 2), (bwd, "backward", 7)):      # QK^T + PV | recomputed S x3, dP, dg, dphi, dtheta
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / iters
        fl = mults * 2.0 * n * L * L * ci
        print(f"N={n} L={L} Ci={ci} {name}: {dt * 1e3:8.2f} ms  {fl / dt / 1e12:6.1f} TFLOP/s executed = {fl / dt / 1e12 / 157.3:.3f} of the 157.3 TF fp32-MFMA peak", flush=True)
    del qkv, y, dy, dqkv
