"""Correctness + throughput probe of the 16-bit-storage contraction kernels (glf_s16_gemm_nt / _tn) through the C ABI.
Run on the GPU box: python profiles/ubench/s16_gemm_probe.py [--perf]"""
import ctypes as C
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from glfusion_amd._lib import GemmParams, check, lib  # noqa: E402

dev = torch.device("cuda:0")
BF = torch.bfloat16


def P(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def S():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def params(M, N, K, lda, ldb, ldc, taps=1, mask=1, tsb=0, gather=0, geo=None, batch=1, bsa=0, bsb=0, bsc=0, alpha=1.0, split=1, rect=0,
           colstats=None, c_bf16=True, ws=None, accumulate=0):
    p = GemmParams()
    p.M, p.N, p.K, p.lda, p.ldb, p.ldc = M, N, K, lda, ldb, ldc
    p.taps, p.tap_mask, p.tap_stride_b, p.gather = taps, mask, tsb, gather
    (p.n_img, p.hs, p.ws, p.hd, p.wd, p.kh, p.kw, p.stride, p.pad, p.dil) = geo if geo else (1, 1, 1, 1, 1, 1, 1, 1, 0, 1)
    p.batch, p.batch_stride_a, p.batch_stride_b, p.batch_stride_c = batch, bsa, bsb, bsc
    p.alpha, p.accumulate, p.split, p.rect = alpha, accumulate, split, rect
    p.colstats = P(colstats)
    p.c_dtype = 1 if c_bf16 else 0
    if ws is not None:
        p.workspace, p.workspace_bytes = P(ws), ws.numel() * 4
    return p


def relerr(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


def conv_case(n, h, w, cin, cout, k, stride, pad, dil, rect=0, stats=False, c_bf16=True, tag=""):
    g = torch.Generator(device="cpu").manual_seed(1)
    x = torch.randn(n, h, w, cin, generator=g).to(dev).to(BF)
    wt = (torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5).to(dev).to(BF)
    ho = (h + 2 * pad - dil * (k - 1) - 1) // stride + 1
    wo = (w + 2 * pad - dil * (k - 1) - 1) // stride + 1
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), wt.float(), None, stride, pad, dil).permute(0, 2, 3, 1).contiguous()
    wtap = wt.permute(2, 3, 0, 1).contiguous().view(k * k, cout, cin)           # [tap][cout][cin]
    y = torch.empty(n, ho, wo, cout, dtype=BF if c_bf16 else torch.float32, device=dev)
    st = torch.zeros(2, cout, dtype=torch.float64, device=dev) if stats else None
    plain = k == 1 and stride == 1 and pad == 0
    from glfusion_amd import ops
    mask = 1 if plain else ops.tap_mask(1, ho, wo, h, w, k, k, stride, pad, dil)
    p = params(n * ho * wo, cout, cin, cin, cin, cout, taps=k * k, mask=mask, tsb=cout * cin, gather=0 if plain else 1,
               geo=None if plain else (n, h, w, ho, wo, k, k, stride, pad, dil), rect=rect, colstats=st, c_bf16=c_bf16)
    check(lib.glf_s16_gemm_nt(P(x), P(wtap), None, P(y), C.byref(p), S()), "nt")
    torch.cuda.synchronize()
    e = relerr(y.float(), ref)
    msg = f"fwd  {tag or ''} n{n} {h}x{w} {cin}->{cout} k{k} s{stride} p{pad} d{dil} rect{rect}: rel {e:.2e}"
    if stats:
        s_ref = ref.double().sum((0, 1, 2)); q_ref = (ref.double() ** 2).sum((0, 1, 2))
        msg += f" stats {relerr(st[0], s_ref):.1e}/{relerr(st[1], q_ref):.1e}"
        assert relerr(st[1], q_ref) < 1e-4, msg
    print(msg)
    assert e < 6e-3, msg
    # dgrad: dx = conv_transpose(dy, w) as NT with gather 2 over [tap][cin][cout]
    dy = torch.randn(n, ho, wo, cout, generator=g).to(dev).to(BF)
    xr = x.float().permute(0, 3, 1, 2).requires_grad_(True)
    wr = wt.float().requires_grad_(True)
    out = F.conv2d(xr, wr, None, stride, pad, dil)
    out.backward(dy.float().permute(0, 3, 1, 2))
    dx_ref = xr.grad.permute(0, 2, 3, 1).contiguous()
    dw_ref = wr.grad                                                         # [cout][cin][k][k]
    wtapT = wt.permute(2, 3, 1, 0).contiguous().view(k * k, cin, cout)       # [tap][cin][cout]
    dx = torch.empty(n, h, w, cin, dtype=BF, device=dev)
    mask2 = 1 if plain else ops.tap_mask(2, h, w, ho, wo, k, k, stride, pad, dil)
    p = params(n * h * w, cin, cout, cout, cout, cin, taps=k * k, mask=mask2, tsb=cout * cin, gather=0 if plain else 2,
               geo=None if plain else (n, ho, wo, h, w, k, k, stride, pad, dil), rect=rect)
    check(lib.glf_s16_gemm_nt(P(dy), P(wtapT), None, P(dx), C.byref(p), S()), "dgrad")
    torch.cuda.synchronize()
    e = relerr(dx.float(), dx_ref)
    print(f"dgrad ...: rel {e:.2e}")
    assert e < 6e-3
    # wgrad: dW_tap[co][ci] = sum_r dy[r][co] x[src(r,tap)][ci]
    rows = n * ho * wo
    for split in (1, 3):
        dwt = torch.empty(k * k, cout, cin, dtype=torch.float32, device=dev).fill_(float("nan"))
        p = params(cout, cin, rows, cout, cin, cin, taps=k * k, mask=mask, tsb=cout * cin, gather=0 if plain else 1,
                   geo=None if plain else (n, h, w, ho, wo, k, k, stride, pad, dil), split=split, c_bf16=False)
        ws = None
        if split > 1:
            nb = int(lib.glf_s16_gemm_tn_workspace_bytes(C.byref(p)))
            ws = torch.empty(nb // 4, dtype=torch.float32, device=dev)
            p.workspace, p.workspace_bytes = P(ws), nb
        if mask != (1 << (k * k)) - 1:
            dwt.zero_()
        check(lib.glf_s16_gemm_tn(P(dy), P(x), P(dwt), C.byref(p), S()), "wgrad")
        torch.cuda.synchronize()
        got = dwt.view(k, k, cout, cin).permute(2, 3, 0, 1)
        e = relerr(got, dw_ref)
        print(f"wgrad split{split}: rel {e:.2e}")
        assert e < 6e-3


def plain_case(M, N, K, batch=1, alpha=1.0, bias=False, c_bf16=True, accumulate=False):
    g = torch.Generator(device="cpu").manual_seed(2)
    A = torch.randn(batch, M, K, generator=g).to(dev).to(BF)
    B = (torch.randn(batch, N, K, generator=g) / K ** 0.5).to(dev).to(BF)
    b = torch.randn(N, generator=g).to(dev) if bias else None
    ref = alpha * torch.bmm(A.float(), B.float().transpose(1, 2)) + (b if bias else 0)
    Cm = torch.empty(batch, M, N, dtype=BF if c_bf16 else torch.float32, device=dev)
    if accumulate:
        Cm.copy_(torch.ones_like(Cm))
        ref = ref + 1
    p = params(M, N, K, K, K, N, batch=batch, bsa=M * K, bsb=N * K, bsc=M * N, alpha=alpha, c_bf16=c_bf16, accumulate=int(accumulate))
    check(lib.glf_s16_gemm_nt(P(A), P(B), P(b), P(Cm), C.byref(p), S()), "nt")
    torch.cuda.synchronize()
    e = relerr(Cm.float(), ref)
    print(f"plain nt M{M} N{N} K{K} b{batch} bf16out={c_bf16} acc={accumulate}: rel {e:.2e}")
    assert e < 6e-3
    # TN: C[m][n] = sum_r A2[r][m] B2[r][n]
    R = K
    A2 = torch.randn(batch, R, M if M % 8 == 0 else 8, generator=g).to(dev).to(BF)
    m2 = A2.shape[2]
    B2 = (torch.randn(batch, R, N if N % 8 == 0 else 8, generator=g) / R ** 0.5).to(dev).to(BF)
    n2 = B2.shape[2]
    ref = alpha * torch.bmm(A2.float().transpose(1, 2), B2.float())
    for c16 in (False, True):
        Ct = torch.empty(batch, m2, n2, dtype=BF if c16 else torch.float32, device=dev)
        p = params(m2, n2, R, m2, n2, n2, batch=batch, bsa=R * m2, bsb=R * n2, bsc=m2 * n2, alpha=alpha, c_bf16=c16)
        check(lib.glf_s16_gemm_tn(P(A2), P(B2), P(Ct), C.byref(p), S()), "tn")
        torch.cuda.synchronize()
        e = relerr(Ct.float(), ref)
        print(f"plain tn M{m2} N{n2} K{R} b{batch} bf16out={c16}: rel {e:.2e}")
        assert e < 6e-3


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def perf():
    print("---- throughput (random data) ----")
    for (M, N, K) in [(150528, 3072, 2048), (50176, 256, 256), (50176, 512, 512), (150528, 2048, 1024), (50176, 2048, 512), (50176, 512, 2048), (50176, 256, 1024), (50176, 1024, 256),
                      (50176, 256, 2048), (193600, 256, 64), (193600, 64, 256), (8192, 8192, 8192)]:
        A = torch.randn(M, K, device=dev).to(BF)
        B = torch.randn(N, K, device=dev).to(BF)
        Cm = torch.empty(M, N, dtype=BF, device=dev)
        p = params(M, N, K, K, K, N)
        ms = timeit(lambda: check(lib.glf_s16_gemm_nt(P(A), P(B), None, P(Cm), C.byref(p), S()), "nt"))
        print(f"NT {M}x{N}x{K}: {ms:.3f} ms  {2.0 * M * N * K / ms / 1e9:.0f} TF  {(M * K + N * K + M * N) * 2 / ms / 1e6:.0f} GB/s")
    for (M, N, R) in [(3072, 2048, 150528), (2048, 1024, 150528), (512, 2048, 50176), (256, 2048, 50176), (256, 1024, 50176), (64, 256, 193600)]:
        A = torch.randn(R, M, device=dev).to(BF)
        B = torch.randn(R, N, device=dev).to(BF)
        Cm = torch.empty(M, N, dtype=torch.float32, device=dev)
        tiles = ((M + 255) // 256) * ((N + 127) // 128)
        split = max(1, min(R // 512, (1024 + tiles - 1) // tiles))
        p = params(M, N, R, M, N, N, split=split, c_bf16=False)
        nb = int(lib.glf_s16_gemm_tn_workspace_bytes(C.byref(p)))
        ws = torch.empty(max(nb // 4, 4), dtype=torch.float32, device=dev)
        if split > 1:
            p.workspace, p.workspace_bytes = P(ws), nb
        ms = timeit(lambda: check(lib.glf_s16_gemm_tn(P(A), P(B), P(Cm), C.byref(p), S()), "tn"))
        print(f"TN {M}x{N}x{R} split {split}: {ms:.3f} ms  {2.0 * M * N * R / ms / 1e9:.0f} TF")
    # a 3x3 conv, layer3-like, and ASPP rate 12 dense vs region
    from glfusion_amd import ops
    for (n, h, cin, cout, dil, rect) in [(64, 28, 256, 256, 2, 0), (64, 28, 512, 512, 4, 0), (64, 28, 2048, 256, 12, 0), (64, 28, 2048, 256, 12, 2),
                                        (64, 28, 2048, 256, 24, 0), (64, 28, 2048, 256, 24, 2)]:
        x = torch.randn(n, h, h, cin, device=dev).to(BF)
        wt = torch.randn(9, cout, cin, device=dev).to(BF)
        y = torch.empty(n, h, h, cout, dtype=BF, device=dev)
        mask = ops.tap_mask(1, h, h, h, h, 3, 3, 1, dil, dil)
        p = params(n * h * h, cout, cin, cin, cin, cout, taps=9, mask=mask, tsb=cout * cin, gather=1, geo=(n, h, h, h, h, 3, 3, 1, dil, dil), rect=rect)
        ms = timeit(lambda: check(lib.glf_s16_gemm_nt(P(x), P(wt), None, P(y), C.byref(p), S()), "nt"))
        print(f"conv3x3 {cin}->{cout} d{dil} rect{rect}: {ms:.3f} ms  dense-equivalent {2.0 * n * h * h * cout * cin * 9 / ms / 1e9:.0f} TF")


def one(kind):
    """a single large launch, ten times: the target of rocprofv3 --pmc runs"""
    M, N, K = 150528, 3072, 2048
    if kind == "nt":
        A = torch.randn(M, K, device=dev).to(BF); B = torch.randn(N, K, device=dev).to(BF); Cm = torch.empty(M, N, dtype=BF, device=dev)
        p = params(M, N, K, K, K, N)
        for _ in range(10):
            check(lib.glf_s16_gemm_nt(P(A), P(B), None, P(Cm), C.byref(p), S()), "nt")
    else:
        A = torch.randn(M, N, device=dev).to(BF); B = torch.randn(M, K, device=dev).to(BF); Cm = torch.empty(N, K, dtype=torch.float32, device=dev)
        p = params(N, K, M, N, K, K, split=4, c_bf16=False)
        nb = int(lib.glf_s16_gemm_tn_workspace_bytes(C.byref(p)))
        ws = torch.empty(nb // 4, dtype=torch.float32, device=dev)
        p.workspace, p.workspace_bytes = P(ws), nb
        for _ in range(10):
            check(lib.glf_s16_gemm_tn(P(A), P(B), P(Cm), C.byref(p), S()), "tn")
    torch.cuda.synchronize()


if __name__ == "__main__":
    torch.manual_seed(0)
    if "--one" in sys.argv:
        one(sys.argv[sys.argv.index("--one") + 1])
        sys.exit(0)
    plain_case(300, 136, 128)
    plain_case(1000, 64, 64, bias=True)
    plain_case(513, 256, 192, batch=3, alpha=0.5, c_bf16=False)
    plain_case(256, 128, 64, accumulate=True)
    plain_case(777, 40, 128, c_bf16=False)
    conv_case(2, 28, 28, 64, 64, 1, 1, 0, 1, stats=True)
    conv_case(2, 28, 28, 64, 128, 3, 1, 1, 1, stats=True)
    conv_case(2, 28, 28, 128, 64, 3, 1, 2, 2)
    conv_case(3, 28, 28, 64, 64, 3, 1, 12, 12)
    conv_case(3, 28, 28, 64, 64, 3, 1, 12, 12, rect=2, stats=True)
    conv_case(3, 28, 28, 64, 192, 3, 1, 24, 24, rect=2)
    conv_case(2, 28, 28, 64, 64, 3, 1, 36, 36)
    conv_case(2, 55, 55, 64, 64, 3, 2, 1, 1)
    conv_case(2, 55, 55, 64, 128, 1, 2, 0, 1)
    conv_case(2, 30, 26, 128, 64, 3, 1, 4, 4, rect=2)
    print("correctness OK")
    if "--perf" in sys.argv:
        perf()
