#!/usr/bin/env python3
"""In-kernel timeline of the split-fp16 NT contraction's main loop (diagnostic builds: `make -C gl-fusion_amd/csrc stamps`
compiles csrc with -DGLF_STAMPS=1 / =2 into lib/libglfusion_stamps.so / libglfusion_stamps2.so, selected through GLF_LIB_PATH;
STAMPS_MODE=2 reads the per-workgroup stamps of the second one, STAMPS_TN=1 times the weight-gradient kernel instead).  One workgroup in the middle of the grid records s_memtime at five
points of 16 consecutive iterations per wave: S0 top, S1 after the first 6 MFMAs were issued, S2 after 12, S3 after 18,
S4 after 24 (before the barrier).  Usage: GLF_LIB_PATH=.../libglfusion_stamps.so stamps.py [precision] [pack]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from glfusion_amd import ops
from glfusion_amd._lib import lib

prec = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
pack = sys.argv[2] if len(sys.argv) > 2 else ""
ops.set_precision(prec)
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
M, N, K = 150528, 3072, 2048
A = torch.rand(M, K, device=dev, generator=g) * 2 - 1
B = torch.rand(N, K, device=dev, generator=g) * 2 - 1
Cm = torch.empty(M, N, device=dev)
ama, amb = ops.amax_of(A), ops.amax_of(B)
pa, pb = "a" in pack, "b" in pack
Ax = ops.packed_of(A, ama) if pa else A
Bx = ops.packed_of(B, amb) if pb else B
CONV = os.environ.get("STAMPS_CONV")               # "cin,cout,dil": a 3x3 conv forward on 64 x 28 x 28 (the gathered NT kernel)
if CONV:
    cin, cout, dil = (int(v) for v in CONV.split(","))
    del A, B, Cm, Ax, Bx
    nimg, hh = 64, 28
    M, N, K = nimg * hh * hh, cout, cin
    A = torch.rand(M, cin, device=dev, generator=g) * 2 - 1
    B = torch.rand(9, cout, cin, device=dev, generator=g) * 2 - 1
    Cm = torch.empty(M, cout, device=dev)
    ama, amb = ops.amax_of(A), ops.amax_of(B)
    Ax = ops.packed_of(A, ama) if pa else A
    Bx = ops.packed_of(B, amb) if pb else B
    geo = (nimg, hh, hh, hh, hh, 3, 3, 1, dil, dil)
    mask = ops.tap_mask(1, hh, hh, hh, hh, 3, 3, 1, dil, dil)
TN = os.environ.get("STAMPS_TN") == "1"            # the weight-gradient kernel instead: C[M][N] = A[K][M]^T B[K][N]
if TN:
    del A, B, Cm, Ax, Bx
    M, N, K = 3072, 2048, 150528
    A = torch.rand(K, M, device=dev, generator=g) * 2 - 1
    B = torch.rand(K, N, device=dev, generator=g) * 2 - 1
    Cm = torch.zeros(M, N, device=dev)
    ama, amb = ops.amax_of(A), ops.amax_of(B)
    Ax = ops.packed_of(A, ama) if pa else A
    Bx = ops.packed_of(B, amb) if pb else B
    sp = ops._tn_split(K, M, N, 1)
for _ in range(40):                                # long enough for the clock to settle
    if CONV:
        ops.gemm("nt", Ax, Bx, Cm, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, taps=9, mask=mask, tap_stride_b=N * K, gather=1, geo=geo,
                 amax_a=ama, amax_b=amb, a_packed=pa, b_packed=pb)
    elif TN:
        ops.gemm("tn", Ax, Bx, Cm, M=M, N=N, K=K, lda=M, ldb=N, ldc=N, split=sp, amax_a=ama, amax_b=amb, a_packed=pa, b_packed=pb)
    else:
        ops.gemm("nt", Ax, Bx, Cm, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, amax_a=ama, amax_b=amb, a_packed=pa, b_packed=pb)
torch.cuda.synchronize()
buf = np.zeros(1024, dtype=np.uint64)
fn = lib.glf_debug_stamps
fn.restype, fn.argtypes = ctypes.c_int, [ctypes.c_void_p]
assert fn(buf.ctypes.data) == 0
if os.environ.get("STAMPS_MODE") == "2":            # workgroup-level stamps (library built with -DGLF_STAMPS=2)
    w = (buf[64:96] if TN else buf[:32]).reshape(8, 4).astype(np.int64)
    nit = int(buf[63]) if TN else (K // 32) * (bin(mask).count("1") if CONV else 1)
    print(f"== {prec} {'tn' if TN else 'nt'} pack={pack or '-'}: one workgroup ({nit} iterations), shader cycles per wave: prologue | main loop (per iteration) | epilogue")
    for i in range(8):
        print(f"{i:4d} {w[i, 1] - w[i, 0]:8d} | {w[i, 2] - w[i, 1]:8d} ({(w[i, 2] - w[i, 1]) / nit:7.1f}) | {w[i, 3] - w[i, 2]:8d}   total {w[i, 3] - w[i, 0]}")
    sys.exit(0)
st = buf.reshape(8, 16, 8)[:, :, :5].astype(np.int64)
print(f"== {prec} pack={pack or '-'}: per-wave iteration period and segment lengths (shader cycles), mean over 15 iterations")
print("wave  period   S0->S1   S1->S2   S2->S3   S3->S4   S4->next S0 (wait + barrier)")
for w in range(8):
    s = st[w]
    per = np.diff(s[:, 0]).mean()
    seg = [(s[:-1, k + 1] - s[:-1, k]).mean() for k in range(4)]
    bar = (s[1:, 0] - s[:-1, 4]).mean()
    print(f"{w:4d} {per:8.0f} " + " ".join(f"{x:8.0f}" for x in seg) + f" {bar:8.0f}")
t0 = st[:, :, 0].min()
print("start of iterations 0..3 per wave, relative to the earliest (skew between waves):")
for w in range(8):
    print(f"{w:4d} " + " ".join(f"{x - t0:7d}" for x in st[w, :4, 0]) + "   S4: " + " ".join(f"{x - t0:7d}" for x in st[w, :4, 4]))
