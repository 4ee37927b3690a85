"""The Global-Local cross-view attention block (TPAVIModule.forward, reference
models/ours.py:845-917) as ONE autograd node over the HIP contraction engine.

x is [N, V, h, w, C] channels-last, i.e. the matrix [N*L, C] with L = V*h*w positions per
frame (what the reference builds with unsqueeze(2)+cat(dim=2), ours.py:1819-1820).

mode 'dot' (the shipped model, ours.py:1746-1747):  f = theta^T phi, y = (f / L) g.  It is linear,
so we re-associate exactly:  M_n = phi_n^T g_n / L  ([Ci,Ci] per frame), y_n = theta_n M_n.  The
[N,L,L] score matrix (1.4 GB at config 2, 983 MB *per frame* at the 5-view 224^2 config) is never
formed and the matmul work drops from 2*L*L*Ci to 2*L*Ci*Ci MACs per frame.
mode 'embedded' (softmax, ours.py:896-897): scores are materialised per frame, normalised by a
row-softmax kernel, then contracted with g.

Tail (ours.py:908-915): w = W_z y + b;  z = LayerNorm_C( BatchNorm3d(w) + x ) in one fused pass.
"""
from __future__ import annotations

import ctypes as C
import os
import weakref

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from ._lib import AttnParams, check, lib
from .ops import (_chk, _contig, _p, _stream, _tn_split, _ws, _wimage, _registry, WJ_COPY, stats_slot, amax_of, amax_slot, colsum, gemm, set_amax, split_mode,
                  tn_needs_zero, transpose2d, weight_T, weight_packed, nt_presplit_ok, tn_presplit_ok, act_packed, packed_hit, pick, zeros, _ones4)


FUSED_SOFTMAX = os.environ.get("GLF_FUSED_SOFTMAX", "1") != "0"
# Exact-fp32 precision only: the block's NT / NN contractions run as K / EXACT_KCHUNK launches that accumulate into C.  The fp32 MFMA
# adds its K products ONE AFTER THE OTHER into the accumulator -- up to a 3 072-long fp32 chain (dx = du + dqkv Wcat) -- and this
# block's gradients cancel to ~1e-3 of their terms: unchunked, theta.weight sat at 4.5e-3 of its norm and the encoder gradients
# behind the block at 2e-3 (the split-fp16 kernels add 16 products per instruction: 5e-4).  Which roundings such a tensor collects is
# a matter of luck at any single chain length (1 024: 1.0e-3 with a dense ASPP forward, 3.6e-3 with per-tap launches; 512: 2.2e-3);
# 256-deep chains were within 1e-3 in every regime measured (5.6e-4 / 9.5e-4), for +14 % on the strict-precision leg
# (profiles/r04_exact_leg_spread.txt -- measurable since the leg is reproducible, see ops.Conv2dFn.forward).
EXACT_KCHUNK = int(os.environ.get("GLF_EXACT_KCHUNK", "256"))
_gemm = gemm


def gemm(mode, A, B, Cm, **kw):          # noqa: F811  (every contraction of this module goes through here)
    K = kw["K"]
    if split_mode() or mode == "tn" or EXACT_KCHUNK <= 0 or K <= EXACT_KCHUNK or kw.get("taps", 1) != 1:
        return _gemm(mode, A, B, Cm, **kw)
    first = True
    for k0 in range(0, K, EXACT_KCHUNK):
        kk = dict(kw)
        kk["K"] = min(EXACT_KCHUNK, K - k0)
        if not first:
            kk["bias"] = None
            kk["accumulate"] = True
        _gemm(mode, A[..., k0:], B[..., k0:] if mode == "nt" else B[..., k0:, :], Cm, **kk)
        first = False
# 'embedded' under the split 16-bit contraction precisions: scores per group of frames through the split-fp16 MFMA kernels (QK^T, PV,
# dP, dS^T theta, dS phi, P^T dY as plain contractions, softmax statistics in fp32, at most CHUNK_BYTES of scores alive) instead of
# the fused exact-fp32 kernel -- 3-4x its rate at the model's shapes; 0 = the fused kernel under every precision
CHUNKED_SOFTMAX = os.environ.get("GLF_CHUNKED_SOFTMAX", "1") != "0"
CHUNK_BYTES = int(os.environ.get("GLF_SOFTMAX_CHUNK_BYTES", str(2 << 30)))


def chunked_softmax_ok(ci: int, L: int) -> bool:
    return CHUNKED_SOFTMAX and split_mode() and ci % 32 == 0 and L % 4 == 0


def _frames_per_chunk(n: int, L: int) -> int:
    lp = (L + 31) // 32 * 32
    return max(1, min(n, CHUNK_BYTES // (L * lp * 4)))


def _scores(th, ph, f0, g, L, lp, ci, c3, am_q, S):
    """S[0:g] = softmax_rows(theta_f phi_f^T) for frames f0 .. f0 + g: rows of stride lp (a multiple of 32: the matrices are K
    operands of the next contractions), padding columns zero."""
    bq = L * c3
    gemm("nt", th[f0 * L:], ph[f0 * L:], S, M=L, N=L, K=ci, lda=c3, ldb=c3, ldc=lp, batch=g, bsa=bq, bsb=bq, bsc=L * lp,
         amax_a=am_q, amax_b=am_q)
    check(lib.glf_softmax_rows_ld(_p(S), g * L, L, lp, _stream()), "softmax_rows_ld")


def _transposed(src, f0, g, L, lp, ci, c3, out):
    """out[0:g] = the [ci, lp] transposes of the [L, ci] column slices src (row stride c3) of frames f0 .. f0 + g, zero beyond L."""
    check(lib.glf_transpose2d_strided(_p(src[f0 * L:]), c3, L * c3, _p(out), lp, ci * lp, L, ci, lp, g, _stream()), "transpose2d_strided")
PACKED_TPAVI = os.environ.get("GLF_PACKED_TPAVI", "1") != "0"      # W_z's output gradient handed over as a packed image (train mode)


def fused_softmax_ok(ci: int) -> bool:
    """The fused QK^T / softmax / PV kernel covers Ci % 32 == 0, Ci <= 1024 (the model: Ci = 1024); other widths (toy
    modules) materialise the scores per frame."""
    return FUSED_SOFTMAX and ci % 32 == 0 and ci <= 1024


def _attn_params(n: int, L: int, ci: int, ldqkv: int, ldy: int) -> AttnParams:
    ap = AttnParams()
    ap.frames, ap.L, ap.ci = n, L, ci
    ap.ldq = ap.ldk = ap.ldv = ldqkv
    ap.ldy, ap.lddy, ap.ldd = ldy, ldy, ldqkv
    return ap


_qkv_cache = {}


def _qkv_weights(params):
    """theta | phi | g weights stacked as ONE [3*ci, c] operand (+ the stacked bias), so the three projections run as a
    single contraction over the shared input.  The stacked buffers live as long as theta's weight; each of the six slices is
    a registered weight image (ops._wimage: re-copied when its source parameter changed, or by ops.refresh_weights), and the
    stacked operand carries the combined version stamp of its sources for the images derived from IT (maximum, transpose,
    packed forms)."""
    th_w, ph_w, g_w, th_b, ph_b, g_b = params
    ci, c = th_w.shape[0], th_w.shape[1]
    if ci % 4 != 0:                                    # odd toy widths only (the 16-byte copy kernel does not apply)
        Wcat = torch.cat([_contig(t.detach()).view(ci, c) for t in (th_w, ph_w, g_w)], dim=0)
        bcat = torch.cat((th_b.detach(), ph_b.detach(), g_b.detach()), dim=0)
        return Wcat, bcat
    key = id(th_w)
    hit = _qkv_cache.get(key)
    if hit is None or hit[0]() is not th_w or hit[1].device != th_w.device or hit[1].shape != (3 * ci, c):
        Wcat = torch.empty(3 * ci, c, dtype=torch.float32, device=th_w.device)
        bcat = torch.empty(3 * ci, dtype=torch.float32, device=th_w.device)
        refs = [weakref.ref(t) for t in params]
        Wcat._glf_version_fn = lambda refs=refs: tuple((r()._version, r().data_ptr()) if r() is not None else None for r in refs)
        Wcat._glf_sources = refs
        hit = _qkv_cache[key] = (weakref.ref(th_w, lambda _r, k=key: _qkv_cache.pop(k, None)), Wcat, bcat)
    _, Wcat, bcat = hit
    for i, (w, b) in enumerate(((th_w, th_b), (ph_w, ph_b), (g_w, g_b))):
        for t, dst, n in ((w, Wcat[i * ci:(i + 1) * ci], ci * c), (b, bcat[i * ci:(i + 1) * ci], ci)):
            src = _contig(t.detach())
            im, fresh = _wimage(t, "qkvcat", WJ_COPY, src, (n, 0, 0), lambda dst=dst: dst)
            if im.dst.data_ptr() != dst.data_ptr():   # the stacked buffers were re-created: re-register
                _registry(t.device).drop(im.key)
                im, fresh = _wimage(t, "qkvcat", WJ_COPY, src, (n, 0, 0), lambda dst=dst: dst)
            if fresh:
                check(lib.glf_copy_frames(_p(src), n, _p(im.dst), n, 1, n, _stream()), "qkv_weights")
    return Wcat, bcat


class TpaviFn(Function):
    @staticmethod
    def forward(ctx, x, th_w, th_b, ph_w, ph_b, g_w, g_b, wz_w, wz_b, bn_g, bn_b, ln_g, ln_b,
                rmean, rvar, nbt, training: bool, momentum: float, bn_eps: float, ln_eps: float, mode: str):
        x = _contig(_chk(x, "TPAVI input"))
        if x.dim() != 5:
            raise RuntimeError("TPAVI input must be [N, V, h, w, C]")
        n, v, h, w_, c = x.shape
        L = v * h * w_
        rows = n * L
        ci = th_w.shape[0]
        dev = x.device
        f32 = dict(dtype=torch.float32, device=dev)
        W = lambda t: _contig(t.detach()).view(t.shape[0], t.shape[1])       # Conv3d 1x1x1 weight -> [out, in]
        zW = W(wz_w)

        # theta | phi | g in ONE contraction over the shared input: qkv[rows, 3*ci] (x is read once; the three
        # operands below are column slices with row stride 3*ci)
        Wcat, bcat = _qkv_weights((th_w, ph_w, g_w, th_b, ph_b, g_b))         # [3*ci, c], [3*ci]
        c3 = 3 * ci
        qkv = torch.empty(rows, c3, **f32)
        am_x = amax_of(x)
        am_q = amax_slot(dev)                # max|qkv| from the epilogue: one bound for the theta | phi | g column slices
        am_wc = amax_of(Wcat)
        ok = nt_presplit_ok(c, c, c)
        wb, pb = pick(Wcat, weight_packed(Wcat, Wcat, "w", am_wc) if ok else None, ok)
        xa, pa = pick(x, act_packed(x, am_x) if ok else None, ok)       # read again by the weight gradient of the projections
        gemm("nt", xa, wb, qkv, M=rows, N=c3, K=c, lda=c, ldb=c, ldc=c3, bias=bcat, amax_a=am_x, amax_b=am_wc, amax_c=am_q,
             a_packed=pa, b_packed=pb)
        ctx.x_packed = (xa, am_x) if (pa and packed_hit(x, am_x) is not None) else None      # retained while memory allows
        ctx.qkv_owner = th_w                      # parameter the stacked operand (and its cached transpose) is keyed on
        set_amax(qkv, am_q)
        th, ph, g = qkv[:, 0:ci], qkv[:, ci:2 * ci], qkv[:, 2 * ci:]
        bq = L * c3                                                          # batch (frame) stride inside qkv

        y = torch.empty(rows, ci, **f32)
        if mode == "dot":
            att = torch.empty(n, ci, ci, **f32)                              # M_n = phi_n^T g_n / L
            am_att = amax_slot(dev)
            gemm("tn", ph, g, att, M=ci, N=ci, K=L, lda=c3, ldb=c3, ldc=ci, batch=n, bsa=bq, bsb=bq,
                 bsc=ci * ci, alpha=1.0 / L, amax_a=am_q, amax_b=am_q, amax_c=am_att)
            set_amax(att, am_att)
            if split_mode() and ci % 32 == 0:      # y_n = theta_n M_n as NT against M_n^T (split-bf16 kernels are NT / TN only)
                attT = transpose2d(att, ci, ci, n)
                am_y = amax_slot(dev)
                gemm("nt", th, attT, y, M=L, N=ci, K=ci, lda=c3, ldb=ci, ldc=ci, batch=n, bsa=bq, bsb=ci * ci, bsc=L * ci,
                     amax_a=am_q, amax_b=am_att, amax_c=am_y)                  # a transpose keeps the maximum
                set_amax(y, am_y)
                del attT
            else:
                gemm("nn", th, att, y, M=L, N=ci, K=ci, lda=c3, ldb=ci, ldc=ci, batch=n, bsa=bq, bsb=ci * ci, bsc=L * ci)
        elif mode == "embedded" and chunked_softmax_ok(ci, L):
            # per group of frames: S = theta phi^T, P = softmax(S) in place, y = P g -- three launches + one transpose of g on the
            # split-fp16 kernels; nothing of size L x L is kept (the backward pass recomputes P group by group)
            att = torch.empty(1, **f32)
            lp = (L + 31) // 32 * 32
            gpc = _frames_per_chunk(n, L)
            S = torch.empty(gpc, L, lp, **f32)
            gT = torch.empty(gpc, ci, lp, **f32)
            one = _ones4(dev)[:1]                                            # max P <= 1
            am_y = amax_slot(dev)
            for f0 in range(0, n, gpc):
                gc = min(gpc, n - f0)
                _scores(th, ph, f0, gc, L, lp, ci, c3, am_q, S)
                _transposed(g, f0, gc, L, lp, ci, c3, gT)
                gemm("nt", S, gT, y[f0 * L:], M=L, N=ci, K=lp, lda=lp, ldb=lp, ldc=ci, batch=gc, bsa=L * lp, bsb=ci * lp, bsc=L * ci,
                     amax_a=one, amax_b=am_q, amax_c=am_y)
            set_amax(y, am_y)
            del S, gT
        elif mode == "embedded" and fused_softmax_ok(ci):
            # ONE kernel: 64-query blocks, key tiles through LDS, online row max / sum, P g in MFMA accumulators; the
            # [L, L] scores are never written.  `att` holds the row log-sum-exp the backward pass recomputes them against.
            att = torch.empty(n * L, **f32)
            ap = _attn_params(n, L, ci, c3, ci)
            check(lib.glf_attn_softmax_fwd(_p(th), _p(ph), _p(g), _p(y), _p(att), C.byref(ap), _stream()), "attn_softmax_fwd")
        elif mode == "embedded":
            att = torch.empty(n, L, L, **f32)                                # softmax(theta phi^T), materialised (odd widths only)
            gemm("nt", th, ph, att, M=L, N=L, K=ci, lda=c3, ldb=c3, ldc=L, batch=n, bsa=bq, bsb=bq, bsc=L * L,
                 amax_a=am_q, amax_b=am_q)
            check(lib.glf_softmax_rows(_p(att), n * L, L, _stream()), "softmax_rows")
            gemm("nn", att, g, y, M=L, N=ci, K=L, lda=L, ldb=c3, ldc=ci, batch=n, bsa=L * L, bsb=bq, bsc=L * ci)
        else:
            raise RuntimeError(f"TPAVI mode {mode!r} is not on the path (built: 'dot', 'embedded')")

        wz = torch.empty(rows, c, **f32)
        am_zw = amax_of(wz_w)
        ok = nt_presplit_ok(ci, ci, ci)
        wb, pb = pick(zW, weight_packed(zW, wz_w, "w", am_zw) if ok else None, ok)
        am_y = amax_of(y)
        ya, pa = pick(y, act_packed(y, am_y) if ok else None, ok)
        # train(): the BatchNorm statistics of w come out of the contraction's own epilogue (column sums of w and w^2 in double,
        # glf_gemm_params.colstats) -- no separate pass over the 1.2 GB tensor
        fuse_stats = training and split_mode() and nt_presplit_ok(ci, ci, ci) and c % 4 == 0
        sums = stats_slot(c, dev) if fuse_stats else None
        gemm("nt", ya, wb, wz, M=rows, N=c, K=ci, lda=ci, ldb=ci, ldc=c, bias=wz_b, amax_a=am_y, amax_b=am_zw, a_packed=pa, b_packed=pb,
             colstats=sums)
        ctx.y_packed = (ya, am_y) if (pa and packed_hit(y, am_y) is not None) else None

        mean = torch.empty(c, **f32)
        invstd = torch.empty(c, **f32)
        if fuse_stats:
            check(lib.glf_bn_stats_from_sums(_p(sums), rows, c, bn_eps, momentum, _p(mean), _p(invstd), _p(rmean), _p(rvar), _p(nbt),
                                             _stream()), "bn_stats_from_sums")
        elif training:
            check(lib.glf_bn_stats(_p(wz), c, rows, c, bn_eps, momentum, _p(mean), _p(invstd), _p(rmean), _p(rvar), _p(nbt),
                                   _p(_ws(rows, c, dev)), _stream()), "bn_stats")
        else:
            check(lib.glf_bn_eval_coeffs(_p(rmean), _p(rvar), bn_eps, _p(mean), _p(invstd), c, _stream()), "bn_eval_coeffs")
        z = torch.empty_like(x)
        rmu = torch.empty(rows, **f32)
        rrs = torch.empty(rows, **f32)
        am_z = amax_slot(dev)                      # max|z|: the heads' first convolutions read z (through the global + local sum)
        check(lib.glf_bn_res_ln_fwd(_p(wz), _p(x), _p(mean), _p(invstd), _p(bn_g), _p(bn_b), _p(ln_g), _p(ln_b), ln_eps,
                                    _p(z), _p(rmu), _p(rrs), rows, c, _p(am_z), _stream()), "bn_res_ln_fwd")
        set_amax(z, am_z)
        ctx.save_for_backward(x, qkv, att, y, wz, mean, invstd, rmu, rrs, Wcat, zW, bn_g, bn_b, ln_g)
        ctx.cfg = (n, L, c, ci, training, mode, tuple(th_w.shape), tuple(wz_w.shape))
        ctx.owners = (wz_w,)                      # parameter owning zW (transposed-copy cache key)
        return z

    @staticmethod
    @once_differentiable
    def backward(ctx, dz):
        (x, qkv, att, y, wz, mean, invstd, rmu, rrs, Wcat, zW, bn_g, bn_b, ln_g) = ctx.saved_tensors
        n, L, c, ci, training, mode, pshape, zshape = ctx.cfg
        rows = n * L
        dev = dz.device
        f32 = dict(dtype=torch.float32, device=dev)
        dz = _contig(dz)
        c3 = 3 * ci
        bq = L * c3
        th, ph, g = qkv[:, 0:ci], qkv[:, ci:2 * ci], qkv[:, 2 * ci:]

        # LayerNorm backward -> du (gradient of u = BN(w) + x); it is also the residual's gradient
        du = torch.empty(rows, c, **f32)
        dln_g = torch.empty(c, **f32)
        dln_b = torch.empty(c, **f32)
        check(lib.glf_bn_res_ln_bwd(_p(dz), _p(wz), _p(x), _p(mean), _p(invstd), _p(bn_g), _p(bn_b), _p(ln_g), _p(rmu), _p(rrs),
                                    _p(du), _p(dln_g), _p(dln_b), rows, c, _p(_ws(rows, c, dev)), _stream()), "bn_res_ln_bwd")
        # BatchNorm3d backward on w.  Its result dwz has three readers: the weight gradient and the dgrad of W_z -- contractions --
        # and W_z's bias gradient, the column sum of dwz.  In train mode that sum is ZERO in exact arithmetic (the bias feeds a
        # BatchNorm: sum_r dwz = -gamma invstd (sum_r xhat) sum(g xhat) / n and sum_r xhat = 0); what fp32 kernels -- the
        # reference's included -- return there is rounding noise.  So in train mode dwz is written ONCE, as the packed image
        # the two contractions read (glf_bn_bwd packed_dx), and the bias gradient is returned as the exact value.
        split = split_mode() and ci % 32 == 0 and c % 32 == 0
        dwz = torch.empty(rows, c, **f32)
        am_dwz_slot = amax_slot(dev)
        dbn_g = torch.empty(c, **f32)
        dbn_b = torch.empty(c, **f32)
        dwz_pk = bool(training and split and PACKED_TPAVI and am_dwz_slot is not None and nt_presplit_ok(c, c, c) and tn_presplit_ok(c, ci, c, ci))
        from .ops import FUSED_BN_BWD, bnbwd_slot
        fused = bnbwd_slot(c, dev) if (FUSED_BN_BWD and c <= 4096) else None
        check(lib.glf_bn_bwd(_p(du), c, _p(wz), c, None, c, _p(mean), _p(invstd), _p(bn_g), None, _p(dwz), c, None, c,
                             _p(dbn_g), _p(dbn_b), rows, c, 0, int(training), None if fused is not None else _p(_ws(rows, c, dev)), _p(am_dwz_slot),
                             int(dwz_pk), None, None, 0, _p(fused), _stream()), "bn_bwd")
        set_amax(dwz, am_dwz_slot)
        # W_z: w = y zW^T + b
        sp = _tn_split(rows, c, ci, 1)
        if not split_mode():
            sp = max(sp, min((rows + 511) // 512, 65535))          # (as for the projections' weight gradient below)
        dzW = (zeros if tn_needs_zero(sp) else torch.empty)(c, ci, **f32)
        am_dwz, am_q = amax_of(dwz), amax_of(qkv)
        ok = tn_presplit_ok(c, ci, c, ci)
        dwz_a, pa = (dwz, True) if dwz_pk else pick(dwz, act_packed(dwz, am_dwz, True) if ok else None, ok)       # shared with the NT contraction below
        am_y = ctx.y_packed[1] if ctx.y_packed is not None else amax_of(y)
        yb, pb = pick(y, ctx.y_packed[0] if ctx.y_packed is not None else None, ok)
        gemm("tn", dwz_a, yb, dzW, M=c, N=ci, K=rows, lda=c, ldb=ci, ldc=ci, split=sp, amax_a=am_dwz, amax_b=am_y, a_packed=pa, b_packed=pb)
        dzb = zeros(c, device=dev) if dwz_pk else colsum(dwz, rows, c)
        (wz_o,) = ctx.owners
        dy = torch.empty(rows, ci, **f32)
        if split:
            am_dy_slot = amax_slot(dev)
            zWT, am_zw = weight_T(zW, wz_o), amax_of(wz_o)
            ok = nt_presplit_ok(c, c, c)
            wb, pb = pick(zWT, weight_packed(zWT, wz_o, "T2", am_zw), ok)
            da, pa = (dwz, True) if dwz_pk else pick(dwz, act_packed(dwz, am_dwz, True) if ok else None, ok)
            gemm("nt", da, wb, dy, M=rows, N=ci, K=c, lda=c, ldb=c, ldc=ci, amax_a=am_dwz, amax_b=am_zw,
                 amax_c=am_dy_slot, a_packed=pa, b_packed=pb)
            del da
            set_amax(dy, am_dy_slot)
        else:
            gemm("nn", dwz, zW, dy, M=rows, N=ci, K=c, lda=c, ldb=ci, ldc=ci)
        del dwz, dwz_a

        dqkv = torch.empty(rows, c3, **f32)                   # [d theta | d phi | d g], row stride 3*ci
        am_dq_slot = amax_slot(dev)                           # its three writers (below) all report into one slot
        dth, dph, dg = dqkv[:, 0:ci], dqkv[:, ci:2 * ci], dqkv[:, 2 * ci:]
        bs = L * ci
        am_dy = amax_of(dy)
        if mode == "dot":
            # y_n = th_n M_n ;  M_n = ph_n^T g_n / L
            gemm("nt", dy, att, dth, M=L, N=ci, K=ci, lda=ci, ldb=ci, ldc=c3, batch=n, bsa=bs, bsb=ci * ci, bsc=bq,
                 amax_a=am_dy, amax_b=amax_of(att), amax_c=am_dq_slot)
            dM = torch.empty(n, ci, ci, **f32)
            am_dM = amax_slot(dev)
            gemm("tn", th, dy, dM, M=ci, N=ci, K=L, lda=c3, ldb=ci, ldc=ci, batch=n, bsa=bq, bsb=bs, bsc=ci * ci,
                 amax_a=am_q, amax_b=am_dy, amax_c=am_dM)
            set_amax(dM, am_dM)
            gemm("nt", g, dM, dph, M=L, N=ci, K=ci, lda=c3, ldb=ci, ldc=c3, batch=n, bsa=bq, bsb=ci * ci, bsc=bq, alpha=1.0 / L,
                 amax_a=am_q, amax_b=am_dM, amax_c=am_dq_slot)
            if split:
                dMT = transpose2d(dM, ci, ci, n)
                gemm("nt", ph, dMT, dg, M=L, N=ci, K=ci, lda=c3, ldb=ci, ldc=c3, batch=n, bsa=bq, bsb=ci * ci, bsc=bq, alpha=1.0 / L,
                     amax_a=am_q, amax_b=am_dM, amax_c=am_dq_slot)
                if am_dq_slot is not None:
                    set_amax(dqkv, am_dq_slot)
                del dMT
            else:
                gemm("nn", ph, dM, dg, M=L, N=ci, K=ci, lda=c3, ldb=ci, ldc=c3, batch=n, bsa=bq, bsb=ci * ci, bsc=bq, alpha=1.0 / L)
        elif chunked_softmax_ok(ci, L):
            # per group of frames: P recomputed; dP = dY g^T; dg = P^T dY; dS = P (dP - rowsum(dP P)); dtheta = dS phi; dphi = dS^T theta
            lp = (L + 31) // 32 * 32
            gpc = _frames_per_chunk(n, L)
            S = torch.empty(gpc, L, lp, **f32)
            dP = torch.empty(gpc, L, lp, **f32)
            phT = torch.empty(gpc, ci, lp, **f32)
            one = _ones4(dev)[:1]
            for f0 in range(0, n, gpc):
                gc = min(gpc, n - f0)
                _scores(th, ph, f0, gc, L, lp, ci, c3, am_q, S)
                am_dP = amax_slot(dev)
                gemm("nt", dy[f0 * L:], g[f0 * L:], dP, M=L, N=L, K=ci, lda=ci, ldb=c3, ldc=lp, batch=gc, bsa=bs, bsb=bq, bsc=L * lp,
                     amax_a=am_dy, amax_b=am_q, amax_c=am_dP)
                gemm("tn", S, dy[f0 * L:], dg[f0 * L:], M=L, N=ci, K=L, lda=lp, ldb=ci, ldc=c3, batch=gc, bsa=L * lp, bsb=bs, bsc=bq,
                     amax_a=one, amax_b=am_dy, amax_c=am_dq_slot)
                check(lib.glf_softmax_rows_bwd_ld(_p(S), _p(dP), gc * L, L, lp, _stream()), "softmax_rows_bwd_ld")       # dP <- dS
                am_dS = amax_slot(dev)                                       # |dS| <= P (|dP| + |sum dP P|) <= 2 max|dP|
                check(lib.glf_amax_combine(_p(am_dP), None, 2.0, 0, _p(am_dS), _stream()), "amax_combine")
                _transposed(ph, f0, gc, L, lp, ci, c3, phT)
                gemm("nt", dP, phT, dth[f0 * L:], M=L, N=ci, K=lp, lda=lp, ldb=lp, ldc=c3, batch=gc, bsa=L * lp, bsb=ci * lp, bsc=bq,
                     amax_a=am_dS, amax_b=am_q, amax_c=am_dq_slot)
                gemm("tn", dP, th[f0 * L:], dph[f0 * L:], M=L, N=ci, K=L, lda=lp, ldb=c3, ldc=c3, batch=gc, bsa=L * lp, bsb=bq, bsc=bq,
                     amax_a=am_dS, amax_b=am_q, amax_c=am_dq_slot)
            if am_dq_slot is not None:
                set_amax(dqkv, am_dq_slot)
            del S, dP, phT
        elif fused_softmax_ok(ci):
            # recompute the score tiles from theta / phi and the saved row log-sum-exp: three passes (dg, dphi, dtheta), each
            # writing its slice of dqkv exactly once
            ap = _attn_params(n, L, ci, c3, ci)
            ap.lddy, ap.ldd = ci, c3
            dsum = torch.empty(rows, **f32)
            check(lib.glf_attn_softmax_bwd(_p(th), _p(ph), _p(g), _p(y), _p(dy), _p(att), _p(dth), _p(dph), _p(dg), _p(dsum), C.byref(ap),
                                           _stream()), "attn_softmax_bwd")
        else:
            # y_n = P_n g_n ; P_n = softmax(th_n ph_n^T)
            dP = torch.empty(n, L, L, **f32)
            gemm("nt", dy, g, dP, M=L, N=L, K=ci, lda=ci, ldb=c3, ldc=L, batch=n, bsa=bs, bsb=bq, bsc=L * L,
                 amax_a=am_dy, amax_b=am_q)
            gemm("tn", att, dy, dg, M=L, N=ci, K=L, lda=L, ldb=ci, ldc=c3, batch=n, bsa=L * L, bsb=bs, bsc=bq,
                 amax_a=amax_of(att), amax_b=am_dy)
            check(lib.glf_softmax_rows_bwd(_p(att), _p(dP), n * L, L, _stream()), "softmax_rows_bwd")   # dP <- dS
            gemm("nn", dP, ph, dth, M=L, N=ci, K=L, lda=L, ldb=c3, ldc=c3, batch=n, bsa=L * L, bsb=bq, bsc=bq)
            gemm("tn", dP, th, dph, M=L, N=ci, K=L, lda=L, ldb=c3, ldc=c3, batch=n, bsa=L * L, bsb=bq, bsc=bq,
                 amax_a=amax_of(dP), amax_b=am_q)
            del dP
        del dy

        # the three projections as one: qkv = x Wcat^T + bcat
        sp = _tn_split(rows, c3, c, 1)
        if not split_mode():
            # exact fp32: this weight gradient's terms cancel to ~1e-3 of their size, and the exact TN kernel adds its K-tile sums in
            # ONE fp32 chain per slice -- slices of at most 512 rows (16 K-tiles; the second stage adds the slabs in double) keep the
            # strict-precision leg at least as accurate as the split-fp16 one (3e-3 -> 1.5e-3 on the smoke fixture)
            sp = max(sp, min((rows + 511) // 512, 65535))
        dWcat = (zeros if tn_needs_zero(sp) else torch.empty)(c3, c, **f32)
        am_dq = amax_of(dqkv)
        ok = tn_presplit_ok(c3, c, c3, c)
        dq_a, pa = pick(dqkv, act_packed(dqkv, am_dq, True) if ok else None, ok)
        am_x = ctx.x_packed[1] if ctx.x_packed is not None else amax_of(x)
        xb, pb = pick(x, ctx.x_packed[0] if ctx.x_packed is not None else None, ok)
        gemm("tn", dq_a, xb, dWcat, M=c3, N=c, K=rows, lda=c3, ldb=c, ldc=c, split=sp, amax_a=am_dq, amax_b=am_x, a_packed=pa, b_packed=pb)
        dbcat = colsum(dqkv, rows, c3)
        grads_w = [dWcat[i * ci:(i + 1) * ci].reshape(pshape) for i in range(3)]
        grads_b = [dbcat[i * ci:(i + 1) * ci] for i in range(3)]
        dx = du                                              # residual gradient, accumulated in place (one RMW)
        if split:
            WcatT = weight_T(Wcat, Wcat)                        # cached with the stacked operand (one rebuild per weight update)
            am_wc = amax_of(Wcat)
            ok = nt_presplit_ok(c3, c3, c3)
            wb, pb = pick(WcatT, weight_packed(WcatT, Wcat, "T2", am_wc), ok)
            da, pa = pick(dqkv, act_packed(dqkv, am_dq, True) if ok else None, ok)
            gemm("nt", da, wb, dx, M=rows, N=c, K=c3, lda=c3, ldb=c3, ldc=c, accumulate=True, amax_a=am_dq, amax_b=am_wc,
                 a_packed=pa, b_packed=pb)
        else:
            gemm("nn", dqkv, Wcat, dx, M=rows, N=c, K=c3, lda=c3, ldb=c, ldc=c, accumulate=True)
        dx = dx.view_as(x)
        return (dx, grads_w[0], grads_b[0], grads_w[1], grads_b[1], grads_w[2], grads_b[2], dzW.view(zshape), dzb,
                dbn_g, dbn_b, dln_g, dln_b, None, None, None, None, None, None, None, None)


def tpavi_forward(x5: torch.Tensor, mod) -> torch.Tensor:
    """x5: [N, V, h, w, C]; mod: a models.ours.TPAVIModule (parameter container)."""
    bn = mod.W_z[1]
    training = bn.training
    if training and bn.momentum is None:
        raise RuntimeError("glfusion_amd: cumulative-average BatchNorm (momentum=None) is not built")
    fn = TpaviFn
    if x5.dtype == torch.bfloat16:
        from .ops16 import Tpavi16Fn
        fn = Tpavi16Fn
    return fn.apply(
        x5, mod.theta.weight, mod.theta.bias, mod.phi.weight, mod.phi.bias, mod.g.weight, mod.g.bias,
        mod.W_z[0].weight, mod.W_z[0].bias, bn.weight, bn.bias, mod.norm_layer.weight, mod.norm_layer.bias,
        bn.running_mean, bn.running_var, bn.num_batches_tracked if training else None,
        training, float(bn.momentum or 0.0), float(bn.eps), float(mod.norm_layer.eps), mod.mode)
