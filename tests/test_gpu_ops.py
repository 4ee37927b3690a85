"""GPU: every HIP kernel (through the C ABI, via glfusion_amd.ops) against a plain PyTorch fp32
CPU reference of the same op on the same seeded inputs.  Tolerances are written per test."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = "cuda"


def rnd(*shape, seed=0, lo=-1.0, hi=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(*shape, generator=g) * (hi - lo) + lo


def close(a, b, tol):
    a = a.detach().cpu().double()
    b = b.detach().cpu().double()
    err = (a - b).abs()
    ok = bool((err <= tol + tol * b.abs()).all())
    if not ok:
        print("max abs err", float(err.max()), "max |ref|", float(b.abs().max()), "tol", tol)
    return ok


@pytest.fixture
def ops(precision):
    from glfusion_amd import ops as _ops
    assert _ops.get_precision() == precision
    return _ops


# ------------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("M,N,K,batch", [(300, 70, 52, 1), (128, 128, 32, 1), (517, 260, 100, 3), (64, 5, 256, 1), (1000, 256, 5, 1),
                                         # <= 64 rows, K % 16 == 0, N >= 64: glf_gemm_nt's skinny kernel (the ASPP pooled branch's shape last)
                                         (1, 64, 16, 1), (37, 131, 80, 1), (64, 256, 2048, 1)])
def test_gemm_nt_nn_tn_plain(ops, M, N, K, batch):
    A = rnd(batch, M, K, seed=1)
    Bnk = rnd(batch, N, K, seed=2)
    bias = rnd(N, seed=3)
    a, b, bi = A.to(DEV), Bnk.to(DEV), bias.to(DEV)
    # NT
    c = torch.empty(batch, M, N, device=DEV)
    ops.gemm("nt", a, b, c, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, bias=bi, batch=batch, bsa=M * K, bsb=N * K, bsc=M * N, alpha=0.5)
    ref = 0.5 * torch.matmul(A.double(), Bnk.double().transpose(1, 2)) + bias.double()
    assert close(c, ref, 5e-5)
    # accumulate
    ops.gemm("nt", a, b, c, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, batch=batch, bsa=M * K, bsb=N * K, bsc=M * N, accumulate=True)
    assert close(c, ref + torch.matmul(A.double(), Bnk.double().transpose(1, 2)), 5e-5)
    # NN
    Bkn = Bnk.transpose(1, 2).contiguous()
    c2 = torch.empty(batch, M, N, device=DEV)
    ops.gemm("nn", a, Bkn.to(DEV), c2, M=M, N=N, K=K, lda=K, ldb=N, ldc=N, batch=batch, bsa=M * K, bsb=N * K, bsc=M * N)
    assert close(c2, torch.matmul(A.double(), Bkn.double()), 5e-5)
    # TN: C[m][n] = sum_r A2[r][m] * B2[r][n], reduction over the M rows here
    B2 = rnd(batch, M, N, seed=4)
    for split in (1, 3):
        c3 = torch.zeros(batch, K, N, device=DEV)
        ops.gemm("tn", a, B2.to(DEV), c3, M=K, N=N, K=M, lda=K, ldb=N, ldc=N, batch=batch, bsa=M * K, bsb=M * N, bsc=K * N, split=split)
        assert close(c3, torch.matmul(A.double().transpose(1, 2), B2.double()), 5e-5)


def test_gemm_bad_arguments_raise(ops):
    a = torch.zeros(4, 4, device=DEV)
    with pytest.raises(RuntimeError, match="M,N,K"):
        ops.gemm("nt", a, a, a, M=0, N=4, K=4, lda=4, ldb=4, ldc=4)
    with pytest.raises(RuntimeError, match="tap_mask"):
        ops.gemm("nt", a, a, a, M=4, N=4, K=4, lda=4, ldb=4, ldc=4, mask=2)


# ------------------------------------------------------------------------------------------ conv
CONVS = [
    # n, h, w, cin, cout, k, stride, pad, dil, bias
    (2, 14, 14, 64, 96, 1, 1, 0, 1, False),
    (2, 15, 13, 32, 40, 3, 1, 1, 1, False),
    (2, 55, 55, 16, 24, 3, 2, 1, 1, False),        # layer2.0.conv2 geometry (55 -> 28)
    (2, 55, 55, 16, 24, 1, 2, 0, 1, False),        # layer2.0.downsample geometry
    (2, 28, 28, 32, 32, 3, 1, 2, 2, False),
    (2, 28, 28, 32, 32, 3, 1, 4, 4, False),
    (2, 28, 28, 64, 32, 3, 1, 12, 12, False),      # ASPP rates on a 28x28 map
    (2, 28, 28, 64, 32, 3, 1, 24, 24, False),
    (2, 28, 28, 64, 32, 3, 1, 36, 36, False),      # centre tap only
    (3, 28, 28, 256, 5, 1, 1, 0, 1, True),         # head output conv (Cout = 5, bias)
    (3, 28, 28, 256, 1, 1, 1, 0, 1, True),         # centerness output conv (Cout = 1, bias)
    (1, 1, 1, 64, 32, 1, 1, 0, 1, False),          # ASPP pooled branch (M = N frames)
    # Cout > 128: the 256-column wgrad tiles of the split kernels (plain, padded gather, rect mode, stride 2)
    (2, 14, 14, 64, 192, 1, 1, 0, 1, False),
    (2, 28, 28, 32, 160, 3, 1, 2, 2, False),
    (2, 28, 28, 64, 288, 3, 1, 12, 12, False),
    (2, 28, 28, 32, 132, 3, 1, 24, 24, False),
    (2, 55, 55, 16, 136, 3, 2, 1, 1, False),
]


@pytest.mark.parametrize("cfg", CONVS)
def test_conv2d_fwd_bwd(ops, cfg):
    n, h, w, cin, cout, k, stride, pad, dil, has_bias = cfg
    x = rnd(n, cin, h, w, seed=10).requires_grad_(True)
    wt = (rnd(cout, cin, k, k, seed=11) / np.sqrt(cin * k * k)).requires_grad_(True)
    b = rnd(cout, seed=12).requires_grad_(True) if has_bias else None
    y_ref = F.conv2d(x, wt, b, stride=stride, padding=pad, dilation=dil)
    gy = rnd(*y_ref.shape, seed=13)
    y_ref.backward(gy)

    xh = x.detach().permute(0, 2, 3, 1).contiguous().to(DEV).requires_grad_(True)
    wd = wt.detach().to(DEV).requires_grad_(True)
    bd = b.detach().to(DEV).requires_grad_(True) if has_bias else None
    y = ops.conv2d(xh, wd, bd, stride, pad, dil)
    assert tuple(y.shape) == (n, y_ref.shape[2], y_ref.shape[3], cout)
    assert close(y.permute(0, 3, 1, 2), y_ref, 2e-5)
    y.backward(gy.permute(0, 2, 3, 1).contiguous().to(DEV))
    assert close(xh.grad.permute(0, 3, 1, 2), x.grad, 5e-5)
    assert close(wd.grad, wt.grad, 1e-4)
    if has_bias:
        assert close(bd.grad, b.grad, 1e-4)


def test_conv1x1_cat(ops):
    xs = [rnd(2, 7, 9, 32, seed=20 + i) for i in range(3)]
    wt = rnd(24, 96, 1, 1, seed=29) / 10
    xr = [t.clone().requires_grad_(True) for t in xs]
    wr = wt.clone().requires_grad_(True)
    ref = F.conv2d(torch.cat([t.permute(0, 3, 1, 2) for t in xr], 1), wr)
    gy = rnd(*ref.shape, seed=30)
    ref.backward(gy)
    xd = [t.to(DEV).requires_grad_(True) for t in xs]
    wd = wt.to(DEV).requires_grad_(True)
    y = ops.conv1x1_cat(wd, xd)
    assert close(y.permute(0, 3, 1, 2), ref, 2e-5)
    y.backward(gy.permute(0, 2, 3, 1).contiguous().to(DEV))
    for a, b in zip(xd, xr):
        assert close(a.grad, b.grad, 5e-5)
    assert close(wd.grad, wr.grad, 1e-4)


def test_stem(ops):
    x = rnd(3, 1, 40, 37, seed=40, lo=0.0)
    wt = (rnd(64, 1, 7, 7, seed=41) / 7).requires_grad_(True)
    b = rnd(64, seed=42).requires_grad_(True)
    ref = F.conv2d(x, wt, b, stride=1, padding=2)
    gy = rnd(*ref.shape, seed=43)
    ref.backward(gy)
    wd, bd = wt.detach().to(DEV).requires_grad_(True), b.detach().to(DEV).requires_grad_(True)
    y = ops.stem7x7(x.permute(0, 2, 3, 1).contiguous().to(DEV), wd, bd, 2)
    assert close(y.permute(0, 3, 1, 2), ref, 1e-5)
    y.backward(gy.permute(0, 2, 3, 1).contiguous().to(DEV))
    assert close(wd.grad, wt.grad, 1e-4)
    assert close(bd.grad, b.grad, 1e-4)


# ------------------------------------------------------------------------------------------ norm
@pytest.mark.parametrize("training,relu,res", [(True, True, False), (True, True, True), (True, False, False), (False, True, True), (False, False, False)])
@pytest.mark.parametrize("shape", [(3, 9, 11, 64), (2, 5, 5, 2048), (4, 1, 1, 256)])
def test_batch_norm_act(ops, training, relu, res, shape):
    c = shape[-1]
    bn = torch.nn.BatchNorm2d(c)
    with torch.no_grad():
        bn.weight.copy_(rnd(c, seed=50, lo=0.5, hi=1.5)); bn.bias.copy_(rnd(c, seed=51))
        bn.running_mean.copy_(rnd(c, seed=52) * 0.1); bn.running_var.copy_(rnd(c, seed=53, lo=0.5, hi=1.5))
    bn.train(training)
    import copy
    bd = copy.deepcopy(bn).to(DEV)
    x = (rnd(*shape, seed=54) * 2 + 0.3)
    r = rnd(*shape, seed=55) if res else None
    xr = x.clone().requires_grad_(True)
    rr = r.clone().requires_grad_(True) if res else None
    ref = bn(xr.permute(0, 3, 1, 2))
    if res:
        ref = ref + rr.permute(0, 3, 1, 2)
    if relu:
        ref = F.relu(ref)
    gy = rnd(*ref.shape, seed=56)
    ref.backward(gy)
    xd = x.to(DEV).requires_grad_(True)
    rd = r.to(DEV).requires_grad_(True) if res else None
    y = ops.batch_norm_act(xd, bd, relu, rd)
    assert close(y.permute(0, 3, 1, 2), ref, 2e-5)
    y.backward(gy.permute(0, 2, 3, 1).contiguous().to(DEV))
    assert close(xd.grad, xr.grad, 1e-4)
    assert close(bd.weight.grad, bn.weight.grad, 2e-4)
    assert close(bd.bias.grad, bn.bias.grad, 2e-4)
    if res:
        assert close(rd.grad, rr.grad, 1e-5)
    assert close(bd.running_mean, bn.running_mean, 1e-6)
    assert close(bd.running_var, bn.running_var, 1e-5)
    assert int(bd.num_batches_tracked) == int(bn.num_batches_tracked)


def test_relu_dropout_maxpool_avgpool(ops):
    x = rnd(2, 13, 11, 64, seed=60)
    xd = x.to(DEV).requires_grad_(True)
    y = ops.relu(xd)
    assert close(y, F.relu(x), 0)
    # maxpool on post-ReLU data (many exact ties at 0): values AND gradient routing must match ATen
    xr = F.relu(x).clone().requires_grad_(True)
    ref = F.max_pool2d(xr.permute(0, 3, 1, 2), 3, 2, 1)
    gy = rnd(*ref.shape, seed=61)
    ref.backward(gy)
    xm = F.relu(x).to(DEV).requires_grad_(True)
    ym = ops.maxpool3x3s2(xm)
    assert close(ym.permute(0, 3, 1, 2), ref, 0)
    ym.backward(gy.permute(0, 2, 3, 1).contiguous().to(DEV))
    assert close(xm.grad, xr.grad, 1e-6)
    # avgpool + broadcast
    xa = x.clone().requires_grad_(True)
    ra = F.adaptive_avg_pool2d(xa.permute(0, 3, 1, 2), 1)
    rb = F.interpolate(ra, size=(13, 11), mode="bilinear", align_corners=False)
    gb = rnd(*rb.shape, seed=62)
    rb.backward(gb)
    xg = x.to(DEV).requires_grad_(True)
    p = ops.global_avgpool(xg)
    assert close(p.permute(0, 3, 1, 2), ra, 1e-6)
    bb = ops.broadcast_hw(p, 13, 11)
    assert close(bb.permute(0, 3, 1, 2), rb, 1e-6)
    bb.backward(gb.permute(0, 2, 3, 1).contiguous().to(DEV))
    assert close(xg.grad, xa.grad, 1e-6)
    # dropout: p = 0.5 keeps ~half, scales by 2, backward uses the same mask
    big = torch.ones(64, 64, 64, device=DEV, requires_grad=True)
    d = ops.dropout(big, 0.5, True)
    keep = (d != 0).float().mean().item()
    assert abs(keep - 0.5) < 0.01
    assert set(torch.unique(d.detach()).cpu().tolist()) == {0.0, 2.0}
    d.sum().backward()
    assert torch.equal(big.grad, d.detach())
    assert ops.dropout(big, 0.5, False) is big and ops.dropout(big, 0.0, True) is big


def test_local_gate(ops):
    n, h, w, c = 2, 6, 5, 128
    cls, ctr, f = rnd(n, h, w, 5, seed=70) * 3, rnd(n, h, w, 1, seed=71) * 3, rnd(n, h, w, c, seed=72)
    cr, tr, fr = cls.clone().requires_grad_(True), ctr.clone().requires_grad_(True), f.clone().requires_grad_(True)
    s = torch.sigmoid(cr.permute(0, 3, 1, 2))
    m = F.adaptive_max_pool3d(s, (1, h, w))                       # the reference's op (ours.py:1805-1807)
    a = torch.sigmoid(20 * m * torch.sigmoid(tr.permute(0, 3, 1, 2)))
    ref = fr.permute(0, 3, 1, 2) * a
    gy = rnd(*ref.shape, seed=73)
    ref.backward(gy)
    cd, td, fd = cls.to(DEV).requires_grad_(True), ctr.to(DEV).requires_grad_(True), f.to(DEV).requires_grad_(True)
    y = ops.local_gate(cd, td, fd, 20)
    assert close(y.permute(0, 3, 1, 2), ref, 1e-5)
    y.backward(gy.permute(0, 2, 3, 1).contiguous().to(DEV))
    assert close(fd.grad, fr.grad, 1e-5)
    assert close(cd.grad, cr.grad, 1e-4)
    assert close(td.grad, tr.grad, 1e-4)


def test_stack_and_add_views(ops):
    xs = [rnd(2, 4, 3, 8, seed=80 + i) for i in range(3)]
    xd = [t.to(DEV).requires_grad_(True) for t in xs]
    st = ops.stack_views(xd)
    assert close(st, torch.stack(xs, 1), 0)
    g, l = rnd(2, 3, 4, 3, 8, seed=85), rnd(2, 3, 4, 3, 8, seed=86)
    gd, ld = g.to(DEV).requires_grad_(True), l.to(DEV).requires_grad_(True)
    outs = ops.add_views(gd, ld)
    for i in range(3):
        assert close(outs[i], g[:, i] + l[:, i], 0)
    gy = [rnd(2, 4, 3, 8, seed=87 + i) for i in range(3)]
    (outs[0] * gy[0].to(DEV)).sum().backward(retain_graph=True)       # only one view used: others get zeros
    assert close(gd.grad[:, 0], gy[0], 0) and float(gd.grad[:, 1:].abs().max()) == 0.0
    (st * torch.stack(gy, 1).to(DEV)).sum().backward()
    for i in range(3):
        assert close(xd[i].grad, gy[i], 0)


@pytest.mark.parametrize("h,w,ho,wo", [(28, 28, 112, 112), (7, 5, 28, 20), (56, 56, 224, 224)])
def test_bilinear_up(ops, h, w, ho, wo):
    x = rnd(2, h, w, 5, seed=90)
    xr = x.clone().requires_grad_(True)
    ref = F.interpolate(xr.permute(0, 3, 1, 2), size=(ho, wo), mode="bilinear", align_corners=False)
    gy = rnd(*ref.shape, seed=91)
    ref.backward(gy)
    xd = x.to(DEV).requires_grad_(True)
    y = ops.bilinear_up(xd, ho, wo)
    assert y.is_contiguous() and tuple(y.shape) == (2, 5, ho, wo)
    assert close(y, ref, 1e-6)
    y.backward(gy.to(DEV))
    assert close(xd.grad, xr.grad, 1e-5)


def test_bce_and_overlap(ops):
    x = rnd(3, 5, 40, 40, seed=100) * 6
    t = (rnd(3, 5, 40, 40, seed=101) > 0.4).float()
    xr = x.clone().requires_grad_(True)
    ref = torch.nn.BCEWithLogitsLoss(reduction="sum")(xr, t)
    (ref * 0.7).backward()
    xd = x.to(DEV).requires_grad_(True)
    loss = ops.bce_with_logits_sum(xd, t.to(DEV))
    assert abs(float(loss) - float(ref)) <= 1e-5 * abs(float(ref))
    (loss * 0.7).backward()
    assert close(xd.grad, xr.grad, 1e-6)
    counts = ops.overlap_counts(xd.detach(), t.to(DEV)).cpu()
    pred = torch.where(torch.sigmoid(x) > 0.5, 1, 0)
    tp = int((pred * t).sum()); fp = int((pred * (1 - t)).sum()); fn = int(((1 - pred) * t).sum()); tn = int(((1 - pred) * (1 - t)).sum())
    assert counts.tolist() == [tp, fp, fn, tn]                        # integer work: bit-exact
    from oracle import glfusion_ref as orc
    want = [float(v) for v in orc.overlap_metrics(t, pred)]
    got = ops.overlap_metrics_from_counts(counts)
    assert np.allclose(got, want, atol=1e-6, rtol=0)


def test_softmax_rows(ops):
    x = rnd(37, 301, seed=110) * 5
    xd = x.to(DEV).clone()
    from glfusion_amd._lib import lib, check
    check(lib.glf_softmax_rows(ops._p(xd), 37, 301, ops._stream()))
    assert close(xd, torch.softmax(x, -1), 1e-6)


@pytest.mark.parametrize("cfg", [(3, 9, 11, 64, 96, 1, 1, 0, 1), (2, 28, 28, 32, 160, 3, 1, 2, 2), (2, 28, 28, 64, 40, 3, 1, 36, 36),
                                 (2, 55, 55, 32, 48, 3, 2, 1, 1), (5, 1, 1, 64, 32, 1, 1, 0, 1)])
def test_conv_epilogue_column_statistics(ops, cfg):
    """f16x3: the conv epilogue accumulates (sum y, sum y^2) per output channel; BatchNorm fed with them must equal
    BatchNorm computing its statistics with its own pass over y."""
    n, h, w, cin, cout, k, stride, pad, dil = cfg
    ops.set_precision("f16x3")
    try:
        x = rnd(n, h, w, cin, seed=21).to(DEV)
        wt = (rnd(cout, cin, k, k, seed=22) / np.sqrt(cin * k * k)).to(DEV)
        assert ops.conv_stats_fusable(wt, stride, pad, dil, h, w)
        sums = torch.zeros(2, cout, dtype=torch.float64, device=DEV)
        y = ops.conv2d(x, wt, None, stride, pad, dil, sums)
        y2 = y.double().reshape(-1, cout)
        assert torch.allclose(sums[0], y2.sum(0), rtol=1e-6, atol=1e-6 * float(y2.abs().sum(0).max()))
        assert torch.allclose(sums[1], (y2 * y2).sum(0), rtol=1e-6)
        bn_a, bn_b = torch.nn.BatchNorm2d(cout).to(DEV).train(), torch.nn.BatchNorm2d(cout).to(DEV).train()
        za = ops.batch_norm_act(y, bn_a, True, None, sums)
        zb = ops.batch_norm_act(y, bn_b, True, None)
        assert torch.allclose(za, zb, rtol=1e-5, atol=1e-5)
        assert torch.allclose(bn_a.running_var, bn_b.running_var, rtol=1e-6) and torch.allclose(bn_a.running_mean, bn_b.running_mean, rtol=1e-5, atol=1e-7)
        # a conv that runs as per-tap rectangles cannot honour it and says so
        wr = (rnd(32, 64, 3, 3, seed=23) / 24.0).to(DEV)
        assert not ops.conv_stats_fusable(wr, 1, 12, 12, 28, 28)
        with pytest.raises(RuntimeError):
            ops.conv2d(rnd(2, 28, 28, 64, seed=24).to(DEV), wr, None, 1, 12, 12, torch.zeros(2, 32, dtype=torch.float64, device=DEV))
        ops.set_precision("f32")
        with pytest.raises(RuntimeError, match="colstats"):
            ops.conv2d(x, wt, None, stride, pad, dil, torch.zeros(2, cout, dtype=torch.float64, device=DEV))
    finally:
        ops.set_precision("f32")


@pytest.mark.parametrize("sa,sb", [(1.0, 1.0), (1e-12, 1e9), (3e7, 1e-3), (1e-30, 1e20)])
def test_f16x3_contraction_accuracy_over_the_fp32_range(sa, sb):
    """The scaled split-fp16 kernels against fp64 for operands anywhere in the fp32 exponent range, with one operand
    row 1e6 below its tensor's maximum (precision is relative to each ELEMENT down to 2^-27 of the operand maximum),
    for caller-supplied maxima (exact, and an 8x loose upper bound) and library-measured ones."""
    from glfusion_amd import ops
    from glfusion_amd._lib import lib
    M, N, K = 512, 256, 4096
    a = rnd(M, K, seed=31) * sa
    b = rnd(N, K, seed=32) * sb
    a[5] *= 1e-6
    ref = a.double() @ b.double().T
    ad, bd = a.to(DEV), b.to(DEV)
    at, bt = a.T.contiguous().to(DEV), b.T.contiguous().to(DEV)        # [K, M], [K, N] for the reduction-over-rows form
    ops.set_precision("f32")
    c32 = torch.empty(M, N, device=DEV)
    ops.gemm("nt", ad, bd, c32, M=M, N=N, K=K, lda=K, ldb=K, ldc=N)
    e32 = float((c32.double().cpu() - ref).abs().max() / ref.abs().max())
    ops.set_precision("f16x3")
    try:
        am = torch.zeros(2, device=DEV)
        lib.glf_amax(ad.data_ptr(), 1, ad.numel(), ad.numel(), am[0:].data_ptr(), None)
        lib.glf_amax(bd.data_ptr(), 1, bd.numel(), bd.numel(), am[1:].data_ptr(), None)
        assert abs(float(am[0]) - float(a.abs().max())) == 0 and abs(float(am[1]) - float(b.abs().max())) == 0
        for maxima in (None, am, am * 8.0):
            kw = {} if maxima is None else dict(amax_a=maxima[0:], amax_b=maxima[1:])
            c = torch.empty(M, N, device=DEV)
            ops.gemm("nt", ad, bd, c, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, **kw)
            err = float((c.double().cpu() - ref).abs().max() / ref.abs().max())
            assert err <= max(3 * e32, 1.5e-6), (err, e32)
            row5 = float((c[5].double().cpu() - ref[5]).abs().max() / ref[5].abs().max())
            assert row5 <= 3e-6, row5                                   # the small row keeps its own relative accuracy
            ct = torch.zeros(M, N, device=DEV)
            ops.gemm("tn", at, bt, ct, M=M, N=N, K=K, lda=M, ldb=N, ldc=N, split=2, **kw)
            errt = float((ct.double().cpu() - ref).abs().max() / ref.abs().max())
            assert errt <= max(3 * e32, 1.5e-6), (errt, e32)
        slot = torch.zeros(1, device=DEV)
        c = torch.empty(M, N, device=DEV)
        ops.gemm("nt", ad, bd, c, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, amax_a=am[0:], amax_b=am[1:], amax_c=slot)
        assert float(slot) == float(c.abs().max())                       # epilogue by-product: max |C|
    finally:
        ops.set_precision("f32")


# ------------------------------------------------------------------------------------------ real model shapes
def _rel_l2(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).norm()) / max(float(b.norm()), 1e-30)


REAL_CONVS = [
    # n, h, w, cin, cout, k, pad, dil      (the contractions that dominate backward at C2: full channel counts)
    (8, 28, 28, 2048, 256, 3, 12, 12),     # ASPP rate 12: per-tap rectangles (fwd / wgrad), dense or region dgrad
    (8, 28, 28, 2048, 256, 3, 24, 24),     # ASPP rate 24
    (8, 28, 28, 2048, 256, 3, 36, 36),     # ASPP rate 36: centre tap only
    (8, 28, 28, 512, 512, 3, 4, 4),        # layer4 conv2
    (192, 28, 28, 1024, 2048, 1, 0, 1),    # W_z of the fusion block at C2: M = 64 frames x 3 views x 784 = 150 528 rows
    (8, 28, 28, 160, 260, 3, 2, 2),        # ragged tiles in every kernel: 260 = 256 + 4 output channels, 160 input channels
]


@pytest.mark.parametrize("cfg", REAL_CONVS)
def test_conv2d_real_shapes_fwd_dgrad_wgrad(ops, cfg):
    """forward, dgrad and wgrad at the 2048-channel model shapes against torch's CPU fp32 convolution (oneDNN), each gated
    at 5e-5 in relative L2 -- the bound that protects the reductions (a wgrad that dropped 0.5 % of its rows is at 5e-3)."""
    n, h, w, cin, cout, k, pad, dil = cfg
    x = rnd(n, cin, h, w, seed=20).requires_grad_(True)
    wt = (rnd(cout, cin, k, k, seed=21) / np.sqrt(cin * k * k)).requires_grad_(True)
    y_ref = F.conv2d(x, wt, None, stride=1, padding=pad, dilation=dil)
    gy = rnd(*y_ref.shape, seed=22)
    y_ref.backward(gy)
    xh = x.detach().permute(0, 2, 3, 1).contiguous().to(DEV).requires_grad_(True)
    wd = wt.detach().to(DEV).requires_grad_(True)
    y = ops.conv2d(xh, wd, None, 1, pad, dil)
    y.backward(gy.permute(0, 2, 3, 1).contiguous().to(DEV))
    assert _rel_l2(y.permute(0, 3, 1, 2), y_ref) <= 5e-5
    assert _rel_l2(xh.grad.permute(0, 3, 1, 2), x.grad) <= 5e-5
    assert _rel_l2(wd.grad, wt.grad) <= 5e-5


@pytest.mark.parametrize("prec", ["f16x3", "f16"])
@pytest.mark.parametrize("cfg", [(3, 28, 28, 64, 96, 1, 1, 0, 1, True), (2, 28, 28, 128, 64, 3, 1, 2, 2, True), (2, 28, 28, 256, 32, 3, 1, 24, 24, False),
                                 (2, 56, 56, 32, 64, 3, 2, 1, 1, True), (2, 28, 28, 512, 64, 3, 1, 12, 12, False),
                                 (4, 28, 28, 256, 512, 1, 1, 0, 1, True),
                                 # ragged tiles: 260 output channels (a 256-wide wgrad tile + a 4-column overhang), 160 input channels
                                 (2, 28, 28, 160, 260, 1, 1, 0, 1, True), (2, 28, 28, 160, 260, 3, 1, 2, 2, True)])
def test_presplit_operands_reproduce_in_kernel_split_bitwise(cfg, prec):
    """Operands handed over in the packed pre-split image (glf_split_f16_packed; A and B of the NT kernels, A and B of the
    TN kernels, plain / gathered / per-tap rectangles / regions) must give the SAME BITS as the same call splitting in its
    staging path: forward, dgrad and wgrad of a conv with the host-side switch on and off.  The convs that run as per-tap
    rectangles add their taps with float atomics in run-to-run order: those are held to 1e-6 of the tensor's maximum."""
    from glfusion_amd import ops as _ops
    n, h, w, cin, cout, k, stride, pad, dil, bitwise = cfg
    x = rnd(n, h, w, cin, seed=70).to(DEV)
    wt = (rnd(cout, cin, k, k, seed=71) / np.sqrt(cin * k * k)).to(DEV)
    keep = (_ops.PRESPLIT, _ops.PRESPLIT_MIN_COLS)
    outs = []
    _ops.set_precision(prec)
    try:
        _ops.PRESPLIT_MIN_COLS = 0              # every eligible operand, whatever the host-side break-even rule says
        for on in (False, True):
            _ops.PRESPLIT = on
            xd, wd = x.clone().requires_grad_(True), wt.clone().requires_grad_(True)
            y = _ops.conv2d(xd, wd, None, stride, pad, dil)
            gy = rnd(*y.shape, seed=72).to(DEV)
            y.backward(gy)
            outs.append((y.detach().clone(), xd.grad.clone(), wd.grad.clone()))
        assert getattr(xd, "_glf_packed", None) is not None, "the pre-split path did not run"
    finally:
        _ops.PRESPLIT, _ops.PRESPLIT_MIN_COLS = keep
        _ops.set_precision("f32")
    for a, b, name in zip(outs[0], outs[1], ("y", "dx", "dw")):
        err = (a - b).abs().max().item()
        if bitwise:
            assert torch.equal(a, b), f"{name}: pre-split differs from in-kernel split by {err:.3e}"
        else:
            assert err <= 1e-6 * a.abs().max().item(), f"{name}: pre-split differs from in-kernel split by {err:.3e}"


def test_presplit_images_are_not_retained_when_memory_is_short():
    """ops.retain_ok: once the allocator has reserved more than PRESPLIT_OFF_FRAC of the device, pre-split images are no
    longer attached to the tensors they were made from (they live for one autograd node) -- and the results do not change."""
    from glfusion_amd import ops as _ops
    n, h, w, cin, cout = 2, 28, 28, 128, 1024
    x = rnd(n, h, w, cin, seed=75).to(DEV)
    wt = (rnd(cout, cin, 1, 1, seed=76) / np.sqrt(cin)).to(DEV)
    gy = rnd(n, h, w, cout, seed=77).to(DEV)
    keep = (_ops.PRESPLIT_OFF_FRAC, dict(_ops._retain_off))
    outs = []
    _ops.set_precision("f16x3")
    try:
        for frac in (keep[0], 0.0):
            _ops.PRESPLIT_OFF_FRAC = frac
            _ops._retain_off.clear()
            _ops.begin_step()                       # the retention decision is taken once per step
            xd, wd = x.clone().requires_grad_(True), wt.clone().requires_grad_(True)
            y = _ops.conv2d(xd, wd, None, 1, 0, 1)
            retained = getattr(xd, "_glf_packed", None) is not None
            assert retained == (frac > 0.0), f"retention {retained} at OFF_FRAC {frac}"
            y.backward(gy)
            outs.append((y.detach().clone(), xd.grad.clone(), wd.grad.clone()))
        assert _ops._retain_off, "the switch did not trip"
    finally:
        _ops.PRESPLIT_OFF_FRAC = keep[0]
        _ops._retain_off.clear()
        _ops._retain_off.update(keep[1])
        _ops.begin_step()
        _ops.set_precision("f32")
    for a, b, name in zip(outs[0], outs[1], ("y", "dx", "dw")):
        assert torch.equal(a, b), f"{name} changed with retention off: {(a - b).abs().max().item():.3e}"


def test_split_f16_packed_layout_and_errors():
    """glf_split_f16_packed: every float4 becomes {h0..h3, l0..l3} with x * s = h + 2^-11 l; bad strides are refused."""
    from glfusion_amd import ops as _ops
    from glfusion_amd._lib import lib
    _ops.set_precision("f16x3")
    try:
        x = (rnd(37, 64, seed=73) * 3.0).to(DEV)
        am = _ops.amax_of(x)
        pk = _ops.packed_of(x, am)
        torch.cuda.synchronize()
        halves = pk.view(torch.float16).view(37, 16, 8).double().cpu()
        x64, amax = x.double().cpu(), float(am.item())
        # the library's scale: the power of two that brings the maximum into [2^13, 2^14)
        recon = None
        for e in range(-40, 40):
            s = 2.0 ** e
            r = ((halves[..., :4] + halves[..., 4:] * 2.0 ** -11) / s).reshape(37, 64)
            if (r - x64).abs().max() <= 2.0 ** -21 * amax:
                recon = e
                break
        assert recon is not None, "no power-of-two scale reconstructs x from the packed image to 2^-21 of its maximum"
        assert lib.glf_split_f16_packed(x.data_ptr(), 37, 62, 64, am.data_ptr(), pk.data_ptr(), 64, None) != 0
        assert lib.glf_split_f16_packed(x.data_ptr(), 37, 64, 62, am.data_ptr(), pk.data_ptr(), 64, None) != 0
        assert lib.glf_split_f16_packed(None, 37, 64, 64, am.data_ptr(), pk.data_ptr(), 64, None) != 0
    finally:
        _ops.set_precision("f32")


@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("relu,res,training", [(True, False, True), (True, True, True), (False, False, True), (True, False, False)])
@pytest.mark.parametrize("shape", [(3, 9, 11, 64), (2, 7, 5, 2048)])
def test_bn_backward_writes_packed_gradient(relu, res, training, shape, fused):
    """glf_bn_bwd(packed_dx = 1): dx comes out as the packed pre-split fp16 image, scaled by an UPPER BOUND of max|dx| that the
    reduction pass derives (max|dy'|, max|xhat|, the two sums per channel).  Checks: the bound is a bound and is tight (< 4x);
    the image reconstructs the fp32 dx to 2^-21 of the bound; dgamma / dbeta / dres are those of the fp32 path.
    fused: the two-launch form (fused_sums: the reduction meets in f64 atomics, the apply kernel finishes the sums and derives the
    bound itself) -- same checks; dgamma / dbeta agree with the three-launch form's to fp32 rounding."""
    from glfusion_amd import ops as _ops
    from glfusion_amd._lib import lib, check
    _ops.set_precision("f16x3")
    try:
        c = shape[-1]
        rows = int(np.prod(shape[:-1]))
        x = (rnd(*shape, seed=81) * 2 + 0.4).to(DEV)
        dy = (rnd(*shape, seed=82) * 3.0).to(DEV)
        gamma, beta = rnd(c, seed=83, lo=0.5, hi=1.5).to(DEV), rnd(c, seed=84).to(DEV)
        mean, invstd = x.view(rows, c).mean(0).contiguous(), (1.0 / (x.view(rows, c).var(0, unbiased=False) + 1e-5).sqrt()).contiguous()
        r = rnd(*shape, seed=85).to(DEV) if res else None
        y = (x - mean) * invstd * gamma + beta
        if res:
            y = y + r
        y = torch.relu(y).contiguous() if relu else y.contiguous()
        outs = []
        for packed in (0, 1):
            dx, dres = torch.empty_like(x), (torch.empty_like(x) if res else None)
            dg, db = torch.empty(c, device=DEV), torch.empty(c, device=DEV)
            am = torch.zeros(1, device=DEV)
            ws = torch.empty(int(lib.glf_bn_workspace(rows, c)), dtype=torch.float64, device=DEV)
            p = lambda t: None if t is None else t.data_ptr()
            fs = torch.zeros(3 * c, dtype=torch.float64, device=DEV) if fused else None
            check(lib.glf_bn_bwd(p(dy), c, p(x), c, p(y) if (relu and res) else None, c, p(mean), p(invstd), p(gamma), p(beta), p(dx), c,
                                 p(dres), c, p(dg), p(db), rows, c, int(relu), int(training), p(ws), p(am), packed, None, None, 0, p(fs), None), "bn_bwd")
            torch.cuda.synchronize()
            outs.append((dx, dres, dg, db, float(am)))
        (dx0, dres0, dg0, db0, am0), (pk, dres1, dg1, db1, bound) = outs
        if fused:       # atomics: the order of the f64 adds is not fixed
            assert torch.allclose(dg0, dg1, rtol=1e-6, atol=1e-6 * float(dg0.abs().max())) and torch.allclose(db0, db1, rtol=1e-6, atol=1e-6 * float(db0.abs().max()))
        else:
            assert torch.equal(dg0, dg1) and torch.equal(db0, db1)
        if res:
            assert torch.equal(dres0, dres1)
        true_max = float(dx0.abs().max())
        assert abs(am0 - true_max) <= 1e-6 * true_max
        assert true_max <= bound <= 4.0 * true_max, (true_max, bound)
        # reconstruct with the library's scale: the power of two that brings the bound into [2^13, 2^14)
        e = int(np.floor(np.log2(bound)))
        s = 2.0 ** (13 - e)
        halves = pk.view(torch.float16).view(rows, c // 4, 8).double().cpu()
        recon = ((halves[..., :4] + halves[..., 4:] * 2.0 ** -11) / s).reshape(rows, c)
        err = float((recon - dx0.view(rows, c).double().cpu()).abs().max())
        assert err <= (2.0 ** -21 + (2e-6 if fused else 0.0)) * bound, (err, bound)
    finally:
        _ops.set_precision("f32")


@pytest.mark.parametrize("packed", [0, 1])
@pytest.mark.parametrize("shape", [(3, 9, 11, 64), (2, 7, 5, 2048)])
def test_bn_backward_adds_two_gradient_addends_while_reading(packed, shape):
    """glf_bn_bwd(dy2): the node sees dy + dy2 without the sum being materialised.  Against the same call on the pre-added
    tensor: dx (fp32 or packed image), dres, dgamma, dbeta and the bound are bit-identical (the kernels add in the same order)."""
    from glfusion_amd import ops as _ops
    from glfusion_amd._lib import lib, check
    _ops.set_precision("f16x3")
    try:
        c = shape[-1]
        rows = int(np.prod(shape[:-1]))
        x = (rnd(*shape, seed=181) * 2 + 0.4).to(DEV)
        a, b = (rnd(*shape, seed=182) * 3.0).to(DEV), (rnd(*shape, seed=186) * 0.7).to(DEV)
        gamma, beta = rnd(c, seed=183, lo=0.5, hi=1.5).to(DEV), rnd(c, seed=184).to(DEV)
        mean, invstd = x.view(rows, c).mean(0).contiguous(), (1.0 / (x.view(rows, c).var(0, unbiased=False) + 1e-5).sqrt()).contiguous()
        r = rnd(*shape, seed=185).to(DEV)
        y = torch.relu((x - mean) * invstd * gamma + beta + r).contiguous()
        p = lambda t: None if t is None else t.data_ptr()
        outs = []
        for dy, dy2 in (((a + b).contiguous(), None), (a, b)):
            dx, dres = torch.empty_like(x), torch.empty_like(x)
            dg, db = torch.empty(c, device=DEV), torch.empty(c, device=DEV)
            am = torch.zeros(1, device=DEV)
            ws = torch.empty(int(lib.glf_bn_workspace(rows, c)), dtype=torch.float64, device=DEV)
            check(lib.glf_bn_bwd(p(dy), c, p(x), c, p(y), c, p(mean), p(invstd), p(gamma), p(beta), p(dx), c, p(dres), c, p(dg), p(db),
                                 rows, c, 1, 1, p(ws), p(am), packed, None, p(dy2), c, None, None), "bn_bwd")
            torch.cuda.synchronize()
            outs.append((dx, dres, dg, db, am))
        for u, v in zip(*outs):
            assert torch.equal(u.view(torch.int32), v.view(torch.int32))
        assert lib.glf_bn_bwd(p(a), c, p(x), c, p(y), c, p(mean), p(invstd), p(gamma), p(beta), p(dx), c, p(dres), c, p(dg), p(db),
                              rows, c, 1, 1, p(ws), p(am), packed, None, p(b), c + 2, None, None) != 0
    finally:
        _ops.set_precision("f32")


def test_stage_hands_block_input_gradients_to_batchnorm_unsummed():
    """models.resnet.Stage: blocks after the first read the previous block's output through fan_out(lazy=True); their two
    gradients reach that block's last BatchNorm as a pair.  Same stage with GLF_LAZY_FAN_IN off (one add_n pass per block):
    input gradient and every parameter gradient agree bit for bit in exact fp32, and the lazy path really ran."""
    from glfusion_amd import ops as _ops
    from glfusion_amd.models.resnet import ResNet
    torch.manual_seed(0)
    net = ResNet(layers=(3, 1, 1, 1))
    stage = net.layer1.to(DEV).train()
    with torch.no_grad():
        for i, prm in enumerate(stage.parameters()):
            prm.copy_(rnd(*prm.shape, seed=300 + i) * (0.2 if prm.dim() > 1 else 1.0) + (1.0 if prm.dim() == 1 and i % 2 == 0 else 0.0))
    x0 = rnd(2, 14, 14, 64, seed=299)
    res, seen = [], []
    orig = _ops.FanOutFn.backward
    for flag in (True, False):
        _ops.LAZY_FAN_IN = flag
        for prm in stage.parameters():
            prm.grad = None
        x = x0.to(DEV).requires_grad_(True)
        y = stage.forward_nhwc(x)
        n_add = [0]
        real_check = _ops.check
        def counting(rc, what, _n=n_add, _c=real_check):
            _n[0] += what == "add_n"
            return _c(rc, what)
        _ops.check = counting
        try:
            y.backward(rnd(*y.shape, seed=298).to(DEV))
        finally:
            _ops.check = real_check
        torch.cuda.synchronize()
        seen.append(n_add[0])
        res.append([x.grad.clone()] + [prm.grad.clone() for prm in stage.parameters()])
    _ops.LAZY_FAN_IN = True
    assert seen == [1, 3], seen            # lazy: only the first block (whose input is not a block output) still sums in a pass
    for a, b in zip(*res):
        assert torch.equal(a, b)


@pytest.mark.parametrize("cfg", [(2, 14, 14, 64, 64, 3, 1, 1, 1), (2, 14, 14, 128, 256, 1, 1, 0, 1), (1, 28, 28, 256, 256, 3, 1, 12, 12),
                                 (2, 15, 15, 64, 128, 3, 2, 1, 1)])
def test_conv_bn_backward_with_packed_gradient_matches_fp32_gradient_path(cfg):
    """conv -> BatchNorm(train) -> ReLU with the BatchNorm gradient handed to the conv as a packed-only image (the default)
    against the same chain with GLF_PACKED_GRADS off (fp32 gradient + split pass): input / weight / BN gradients agree to
    fp32-noise level (the two differ only in the power-of-two scale of one operand)."""
    from glfusion_amd import ops as _ops
    from glfusion_amd.models.layers import BatchNorm2d, Conv2d, conv_bn_act
    n, h, w, cin, cout, k, stride, pad, dil = cfg
    _ops.set_precision("f16x3")
    try:
        conv = Conv2d(cin, cout, k, stride=stride, padding=pad, dilation=dil, bias=False)
        bn = BatchNorm2d(cout)
        with torch.no_grad():
            conv.weight.copy_(rnd(*conv.weight.shape, seed=91) * 0.2)
            bn.weight.copy_(rnd(cout, seed=92, lo=0.5, hi=1.5)); bn.bias.copy_(rnd(cout, seed=93))
        conv, bn = conv.to(DEV), bn.to(DEV).train()
        x0 = rnd(n, h, w, cin, seed=94)
        res = []
        saw_packed = []
        for flag in (True, False):
            _ops.PACKED_GRADS = flag
            for p in list(conv.parameters()) + list(bn.parameters()):
                p.grad = None
            x = x0.to(DEV).requires_grad_(True)
            y = conv_bn_act(x, conv, bn, relu=True)
            gy = rnd(*y.shape, seed=95).to(DEV)
            seen = []
            hook = y.grad_fn.next_functions[0][0].register_prehook(lambda g: seen.append(_ops.packed_only(g[0]))) if flag else None
            y.backward(gy)
            torch.cuda.synchronize()
            res.append((x.grad.clone(), conv.weight.grad.clone(), bn.weight.grad.clone(), bn.bias.grad.clone()))
            saw_packed.append(seen)
        assert saw_packed[0] == [True], "the conv did not receive a packed-only gradient"
        for a, b, name in zip(res[0], res[1], ("dx", "dw", "dgamma", "dbeta")):
            err = float((a - b).norm() / (b.norm() + 1e-30))
            assert err <= 2e-6, (name, err)
    finally:
        _ops.PACKED_GRADS = True
        _ops.set_precision("f32")


def test_colmax_is_refused_where_no_epilogue_folds_it():
    """glf_gemm_params.colmax exists in the wide-store epilogue of the split-fp16 NT kernels only; a call that would run any other
    epilogue (exact fp32 kernels, an unaligned C) fails with GLF_ERR_UNSUPPORTED instead of leaving the maxima at zero."""
    from glfusion_amd import ops as _ops
    rows, c, k = 300, 64, 64
    a, b = rnd(rows, k, seed=1).to(DEV), rnd(c, k, seed=2).to(DEV)
    for precision, ldc, ok in (("f16x3", c, True), ("f32", c, False), ("f16x3", c + 2, False)):
        _ops.set_precision(precision)
        try:
            x = torch.empty(rows, ldc, device=DEV)
            sums, colmax = _ops.stats_slot(c, DEV), _ops.colmax_slot(c, DEV)
            if ok:
                _ops.gemm("nt", a, b, x, M=rows, N=c, K=k, lda=k, ldb=k, ldc=ldc, colstats=sums, colmax=colmax)
            else:
                with pytest.raises(RuntimeError, match="colmax|colstats"):
                    _ops.gemm("nt", a, b, x, M=rows, N=c, K=k, lda=k, ldb=k, ldc=ldc, colstats=sums, colmax=colmax)
        finally:
            _ops.set_precision("f32")


@pytest.mark.parametrize("relu", [True, False])
@pytest.mark.parametrize("shape,k", [((3, 9, 11, 64), 64), ((2, 7, 5, 512), 256), ((1, 33, 17, 128), 1152)])
def test_bn_forward_writes_packed_activation(relu, shape, k):
    """glf_gemm_nt(colmax) + glf_bn_apply_from_sums(colmax): the contraction's epilogue leaves the per-channel maxima of |x|
    next to the batch sums, and the BatchNorm apply writes y directly as the packed pre-split fp16 image, scaled by a bound of
    max|y| known before y exists.  Checks: colmax is exactly the column maxima of the stored C; the bound is a bound and is
    tight (< 4x); the image reconstructs the fp32 y to 2^-21 of the bound; mean / invstd / running statistics are those of the
    fp32-output call."""
    from glfusion_amd import ops as _ops
    from glfusion_amd._lib import lib, check
    _ops.set_precision("f16x3")
    try:
        c = shape[-1]
        rows = int(np.prod(shape[:-1]))
        a = (rnd(rows, k, seed=61) + 0.1).to(DEV)
        b = (rnd(c, k, seed=62) / np.sqrt(k)).to(DEV)
        x = torch.empty(rows, c, device=DEV)
        sums = _ops.stats_slot(c, DEV)
        colmax = _ops.colmax_slot(c, DEV)
        _ops.gemm("nt", a, b, x, M=rows, N=c, K=k, lda=k, ldb=k, ldc=c, colstats=sums, colmax=colmax)
        torch.cuda.synchronize()
        assert torch.equal(colmax, x.abs().amax(dim=0)), "colmax is not the column maxima of the stored C"
        gamma, beta = rnd(c, seed=63, lo=0.5, hi=1.5).to(DEV), rnd(c, seed=64).to(DEV)
        p = lambda t: None if t is None else t.data_ptr()
        outs = []
        for packed in (False, True):
            y = torch.empty_like(x)
            mean, invstd = torch.empty(c, device=DEV), torch.empty(c, device=DEV)
            rm, rv, nbt = torch.zeros(c, device=DEV), torch.ones(c, device=DEV), torch.zeros((), dtype=torch.int64, device=DEV)
            am = torch.zeros(1, device=DEV)
            check(lib.glf_bn_apply_from_sums(p(x), c, None, c, p(y), c, p(sums), rows, c, 1e-5, 0.1, p(gamma), p(beta), p(mean), p(invstd),
                                             p(rm), p(rv), p(nbt), int(relu), p(am), None, p(colmax) if packed else None, None), "bn_apply_from_sums")
            torch.cuda.synchronize()
            outs.append((y, mean, invstd, rm, rv, int(nbt), float(am)))
        (y0, m0, i0, rm0, rv0, n0, am0), (pk, m1, i1, rm1, rv1, n1, bound) = outs
        assert torch.equal(m0, m1) and torch.equal(i0, i1) and torch.equal(rm0, rm1) and torch.equal(rv0, rv1) and n0 == n1 == 1
        true_max = float(y0.abs().max())
        assert abs(am0 - true_max) <= 1e-6 * true_max
        assert true_max <= bound <= 4.0 * true_max, (true_max, bound)
        e = int(np.floor(np.log2(bound)))
        s = 2.0 ** (13 - e)
        halves = pk.view(torch.float16).view(rows, c // 4, 8).double().cpu()
        recon = ((halves[..., :4] + halves[..., 4:] * 2.0 ** -11) / s).reshape(rows, c)
        err = float((recon - y0.double().cpu()).abs().max())
        assert err <= 2.0 ** -21 * bound, (err, bound)
        # refused: a residual, in place, or no amax slot with the packed output
        assert lib.glf_bn_apply_from_sums(p(x), c, p(y0), c, p(pk), c, p(sums), rows, c, 1e-5, 0.1, p(gamma), p(beta), p(m1), p(i1),
                                          None, None, None, int(relu), p(am), None, p(colmax), None) != 0
        assert lib.glf_bn_apply_from_sums(p(x), c, None, c, p(x), c, p(sums), rows, c, 1e-5, 0.1, p(gamma), p(beta), p(m1), p(i1),
                                          None, None, None, int(relu), p(am), None, p(colmax), None) != 0
        assert lib.glf_bn_apply_from_sums(p(x), c, None, c, p(pk), c, p(sums), rows, c, 1e-5, 0.1, p(gamma), p(beta), p(m1), p(i1),
                                          None, None, None, int(relu), None, None, p(colmax), None) != 0
    finally:
        _ops.set_precision("f32")


@pytest.mark.parametrize("cfg", [(2, 14, 14, 256, 64, 3, 1, 1, 1), (2, 15, 15, 128, 128, 3, 2, 1, 1), (1, 32, 32, 1024, 256, 3, 1, 2, 2),
                                 (2, 14, 14, 64, 64, 1, 1, 0, 1)])
def test_bottleneck_inner_activation_as_packed_image_matches_fp32_activation_path(cfg):
    """conv1 -> BN -> ReLU -> conv2 -> BN -> ReLU with the activation between the convs written ONLY as the packed image conv2
    reads (the default inside a bottleneck) against the same chain with GLF_PACKED_ACTS off (fp32 activation + split pass):
    the output and every gradient agree to fp32-noise level (the two differ only in the power-of-two scale of one operand)."""
    from glfusion_amd import ops as _ops
    from glfusion_amd.models.layers import BatchNorm2d, Conv2d, conv_bn_act
    n, h, w, cin, mid, k, stride, pad, dil = cfg
    _ops.set_precision("f16x3")
    try:
        conv1, bn1 = Conv2d(cin, mid, 1, bias=False), BatchNorm2d(mid)
        conv2, bn2 = Conv2d(mid, mid, k, stride=stride, padding=pad, dilation=dil, bias=False), BatchNorm2d(mid)
        with torch.no_grad():
            conv1.weight.copy_(rnd(*conv1.weight.shape, seed=101) * 0.2)
            conv2.weight.copy_(rnd(*conv2.weight.shape, seed=102) * 0.2)
            for i, bn in enumerate((bn1, bn2)):
                bn.weight.copy_(rnd(mid, seed=103 + i, lo=0.5, hi=1.5)); bn.bias.copy_(rnd(mid, seed=105 + i))
        mods = [m.to(DEV).train() for m in (conv1, bn1, conv2, bn2)]
        params = [p for m in mods for p in m.parameters()]
        x0 = rnd(n, h, w, cin, seed=107)
        res, saw = [], []
        for flag in (True, False):
            _ops.PACKED_ACTS = flag
            for p in params:
                p.grad = None
            x = x0.to(DEV).requires_grad_(True)
            mid_act = conv_bn_act(x, conv1, bn1, relu=True, consumer=conv2)
            saw.append(_ops.packed_only(mid_act))
            y = conv_bn_act(mid_act, conv2, bn2, relu=True)
            y.backward(rnd(*y.shape, seed=108).to(DEV))
            torch.cuda.synchronize()
            res.append([y.detach().clone(), x.grad.clone()] + [p.grad.clone() for p in params])
        assert saw == [True, False], saw
        for i, (a, b) in enumerate(zip(res[0], res[1])):
            err = float((a - b).norm() / (b.norm() + 1e-30))
            assert err <= 2e-6, (i, err)
    finally:
        _ops.PACKED_ACTS = True
        _ops.set_precision("f32")


@pytest.mark.parametrize("shape", [(3, 112, 112), (2, 57, 40), (1, 9, 9), (2, 224, 224)])
def test_fused_inference_stem_equals_the_three_kernel_chain(shape):
    """glf_stem7x7_bn_relu_pool: conv7x7 + bias -> BatchNorm (running statistics) -> ReLU -> max-pool 3x3/2 in one launch
    against the conv / BatchNorm / pool kernels it replaces under no_grad evaluation: bit-identical values and maximum; and against
    torch's own ops on the CPU at 1e-5.  Odd sizes exercise ragged pooled tiles and windows that hang over the conv output."""
    from glfusion_amd import ops as _ops
    from glfusion_amd.models.layers import BatchNorm2d, Conv2d, MaxPool2d, conv_bn_act, init_block_nhwc
    n, h, w = shape
    conv, bn, pool = Conv2d(1, 64, kernel_size=7, stride=1, padding=2), BatchNorm2d(64), MaxPool2d(kernel_size=3, stride=2, padding=1)
    with torch.no_grad():
        conv.weight.copy_(rnd(64, 1, 7, 7, seed=401) * 0.3); conv.bias.copy_(rnd(64, seed=402) * 0.2)
        bn.weight.copy_(rnd(64, seed=403, lo=0.5, hi=1.5)); bn.bias.copy_(rnd(64, seed=404) * 0.5)
        bn.running_mean.copy_(rnd(64, seed=405) * 0.3); bn.running_var.copy_(rnd(64, seed=406, lo=0.5, hi=2.0))
    x = rnd(n, 1, h, w, seed=407)
    want = F.max_pool2d(torch.relu(F.batch_norm(F.conv2d(x, conv.weight, conv.bias, 1, 2), bn.running_mean, bn.running_var, bn.weight, bn.bias,
                                                 False, 0.1, bn.eps)), 3, 2, 1).detach()
    conv, bn = conv.to(DEV).eval(), bn.to(DEV).eval()
    xd = _ops.to_nhwc(x.to(DEV))
    for prec in ("f32", "f16x3"):
        _ops.set_precision(prec)
        try:
            with torch.no_grad():
                fused = init_block_nhwc(xd, conv, bn, pool)
                _ops.FUSED_STEM = False
                chain = init_block_nhwc(xd, conv, bn, pool)
                _ops.FUSED_STEM = True
            torch.cuda.synchronize()
            assert fused.shape == chain.shape == (n, (h - 2 - 1) // 2 + 1, (w - 2 - 1) // 2 + 1, 64)
            assert torch.equal(fused, chain)
            np.testing.assert_allclose(fused.permute(0, 3, 1, 2).cpu().numpy(), want.numpy(), rtol=1e-5, atol=1e-5)
            if prec == "f16x3":
                assert float(_ops.amax_of(fused)) == float(fused.abs().max())
        finally:
            _ops.FUSED_STEM = True
            _ops.set_precision("f32")
    # training mode and grad mode keep the differentiable chain
    bn.train()
    with torch.no_grad():
        assert init_block_nhwc(xd, conv, bn, pool).shape == fused.shape
    from glfusion_amd._lib import lib
    assert lib.glf_stem7x7_bn_relu_pool(xd.data_ptr(), conv.weight.data_ptr(), None, bn.running_mean.data_ptr(), bn.running_var.data_ptr(),
                                        bn.weight.data_ptr(), bn.bias.data_ptr(), fused.data_ptr(), n, h, w, 32, 2, None, None) != 0
    assert lib.glf_stem7x7_bn_relu_pool(None, conv.weight.data_ptr(), None, bn.running_mean.data_ptr(), bn.running_var.data_ptr(),
                                        bn.weight.data_ptr(), bn.bias.data_ptr(), fused.data_ptr(), n, h, w, 64, 2, None, None) != 0


def test_mfma_probe_counts_what_it_claims():
    """glf_probe_mfma_f16 (bench.py's in-run power-limited peak): with constant operands (seed 0: all ones) every accumulator
    element grows by K = 16 per MFMA, so each thread stores 16 elements x 4 accumulators x 16 x iters -- the launch really
    issued blocks x 8 x iters x 4 MFMAs.  Bad arguments are refused.  The probe lives in the diagnostic library
    (include/glfusion_diag.h), not in the product one."""
    import ctypes
    from glfusion_amd import _lib
    lib = ctypes.CDLL(os.path.join(os.path.dirname(_lib.LIB_PATH), "libglfusion_diag.so"))
    lib.glf_probe_mfma_f16.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_uint32, ctypes.c_void_p]
    assert not hasattr(ctypes.CDLL(_lib.LIB_PATH), "glf_probe_mfma_f16")
    blocks, iters = 8, 10
    out = torch.zeros(blocks * 512, device=DEV)
    assert lib.glf_probe_mfma_f16(out.data_ptr(), blocks, iters, 0, None) == 0
    torch.cuda.synchronize()
    assert torch.equal(out, torch.full_like(out, 16.0 * 4 * 16 * iters))
    assert lib.glf_probe_mfma_f16(out.data_ptr(), blocks, iters, 7, None) == 0
    torch.cuda.synchronize()
    assert bool(torch.isfinite(out).all()) and float(out.abs().max()) > 0
    assert lib.glf_probe_mfma_f16(None, blocks, iters, 0, None) != 0
    assert lib.glf_probe_mfma_f16(out.data_ptr(), 0, iters, 0, None) != 0
    assert lib.glf_probe_mfma_f16(out.data_ptr(), blocks, 0, 0, None) != 0


# ------------------------------------------------------------------------------------------ f16x3 range adversaries
def test_f16x3_outlier_and_small_view_operands():
    """The split-fp16 kernels scale each operand by ONE power of two taken from its maximum.  Adversaries: (a) a gradient
    tensor with a single 1e6 outlier (every other element sits 2^20 below the scale), (b) a stacked activation tensor in
    which one view is 2^30 smaller than the others.  Results must stay at fp32 level: 1e-5 relative L2 per output slice
    against fp64 for (a); for (b) the small view's rows keep >= 19 bits (error relative to THAT view's magnitude <= 1e-5)."""
    from glfusion_amd import ops as _ops
    _ops.set_precision("f16x3")
    try:
        n, h, w, cin, cout = 2, 28, 28, 256, 256
        x = rnd(n, h, w, cin, seed=30)
        wt = rnd(cout, cin, 3, 3, seed=31) / np.sqrt(cin * 9)
        gy = rnd(n, h, w, cout, seed=32)
        r0, c0 = 777, 13
        gy.view(-1, cout)[r0, c0] = 1.0e6
        xd = x.to(DEV).requires_grad_(True)
        wd = wt.to(DEV).requires_grad_(True)
        _ops.conv2d(xd, wd, None, 1, 2, 2).backward(gy.to(DEV))
        x64 = x.double().permute(0, 3, 1, 2).requires_grad_(True)
        w64 = wt.double().requires_grad_(True)
        F.conv2d(x64, w64, None, 1, 2, 2).backward(gy.double().permute(0, 3, 1, 2))
        dx_ref = x64.grad.permute(0, 2, 3, 1)
        # dgrad rows that the outlier touches (its 3x3 dilated footprint) and all the others, separately
        touched = (dx_ref.abs().amax(dim=3) > 1e3)
        assert 1 <= int(touched.sum()) <= 9
        assert _rel_l2(xd.grad[touched.to(DEV)], dx_ref[touched]) <= 1e-5
        assert _rel_l2(xd.grad[(~touched).to(DEV)], dx_ref[~touched]) <= 1e-5
        # wgrad: the outlier's output channel and all the others, separately
        others = [c for c in range(cout) if c != c0]
        assert _rel_l2(wd.grad[c0], w64.grad[c0]) <= 1e-5
        assert _rel_l2(wd.grad[others], w64.grad[others]) <= 1e-5

        # (b) rows of one view 2^30 below the rest inside one operand (one amax for the whole tensor)
        rows, k, nn_ = 3 * 784, 512, 256
        a = rnd(rows, k, seed=33)
        a[784:1568] *= 2.0 ** -30
        b = rnd(nn_, k, seed=34) / np.sqrt(k)
        c = torch.empty(rows, nn_, device=DEV)
        _ops.gemm("nt", a.to(DEV), b.to(DEV), c, M=rows, N=nn_, K=k, lda=k, ldb=k, ldc=nn_)
        ref = a.double() @ b.double().t()
        assert _rel_l2(c[:784], ref[:784]) <= 2e-6 and _rel_l2(c[1568:], ref[1568:]) <= 2e-6
        assert _rel_l2(c[784:1568], ref[784:1568]) <= 1e-5          # 2^-49 amax absolute floor = 2^-19 of this view
    finally:
        _ops.set_precision("f32")


# ------------------------------------------------------------------------------------------ fused softmax attention
def _attn_ref(th, ph, g, dy):
    th, ph, g = (t.double().requires_grad_(True) for t in (th, ph, g))
    y = torch.softmax(th @ ph.transpose(1, 2), dim=-1) @ g
    y.backward(dy.double())
    return y.detach(), th.grad, ph.grad, g.grad


@pytest.mark.parametrize("n,L,ci,scale", [(2, 90, 32, 1.0), (1, 200, 64, 0.5), (3, 64, 128, 0.3), (2, 2352, 1024, 0.05), (1, 15680, 1024, 0.05)])
def test_fused_softmax_attention_fwd_bwd(n, L, ci, scale):
    """glf_attn_softmax_{fwd,bwd} (ours.py:881, 896-897, 902 in one kernel per pass) against softmax(theta phi^T) g in
    fp64 on the host: y and the three gradients within 2e-5 relative L2, on ragged L (90, 200: partial 64-row blocks),
    the config-2 length 2352 and the config-5 length 15 680, operands read as column slices of one [rows, 3 Ci] buffer."""
    import ctypes as C
    from glfusion_amd._lib import AttnParams, check, lib
    qkv = rnd(n, L, 3 * ci, seed=40) * scale
    dy = rnd(n, L, ci, seed=41)
    th, ph, g = qkv[..., :ci], qkv[..., ci:2 * ci], qkv[..., 2 * ci:]
    y_ref, dth_ref, dph_ref, dg_ref = _attn_ref(th, ph, g, dy)
    d = qkv.to(DEV)
    dyd = dy.to(DEV)
    y = torch.empty(n * L, ci, device=DEV)
    lse = torch.empty(n * L, device=DEV)
    dqkv = torch.full((n * L, 3 * ci), float("nan"), device=DEV)          # every element must be written
    dsum = torch.empty(n * L, device=DEV)
    ap = AttnParams()
    ap.frames, ap.L, ap.ci = n, L, ci
    ap.ldq = ap.ldk = ap.ldv = 3 * ci
    ap.ldy, ap.lddy, ap.ldd = ci, ci, 3 * ci
    p = lambda t, off=0: C.c_void_p(t.data_ptr() + 4 * off)
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    check(lib.glf_attn_softmax_fwd(p(d), p(d, ci), p(d, 2 * ci), p(y), p(lse), C.byref(ap), s), "attn fwd")
    check(lib.glf_attn_softmax_bwd(p(d), p(d, ci), p(d, 2 * ci), p(y), p(dyd), p(lse), p(dqkv), p(dqkv, ci), p(dqkv, 2 * ci), p(dsum),
                                   C.byref(ap), s), "attn bwd")
    torch.cuda.synchronize()
    lse_ref = torch.logsumexp(th.double() @ ph.double().transpose(1, 2), dim=-1).reshape(-1)
    assert float((lse.cpu().double() - lse_ref).abs().max()) <= 1e-5 * max(1.0, float(lse_ref.abs().max()))
    assert _rel_l2(y.view(n, L, ci), y_ref) <= 2e-5
    dq = dqkv.view(n, L, 3 * ci)
    assert bool(torch.isfinite(dq).all())
    assert _rel_l2(dq[..., 2 * ci:], dg_ref) <= 2e-5
    assert _rel_l2(dq[..., ci:2 * ci], dph_ref) <= 2e-5
    assert _rel_l2(dq[..., :ci], dth_ref) <= 2e-5


def test_fused_softmax_matches_materialised_path():
    """TPAVIModule(mode='embedded') through the fused kernels vs the same module with the scores materialised per frame
    (three launches + row softmax) at the config-2 length L = 3 x 28 x 28 = 2352: outputs and every gradient agree."""
    from glfusion_amd import fusion
    from glfusion_amd.models import TPAVIModule
    from oracle import glfusion_ref as orc
    res = {}
    for fused in (True, False):
        fusion.FUSED_SOFTMAX = fused
        try:
            m = TPAVIModule(256, mode="embedded")
            orc.closed_form_fill(m, salt=3)
            m = m.to(DEV).train()
            x = (orc.closed_form_tensor((2, 256, 3, 28, 28), 101, -1.0, 1.0) * 0.5).to(DEV).requires_grad_(True)
            z, _ = m(x)
            (z * orc.closed_form_tensor(tuple(z.shape), 102, -1.0, 1.0).to(DEV)).sum().backward()
            res[fused] = (z.detach(), x.grad, {k: p.grad for k, p in m.named_parameters() if p.grad is not None})
        finally:
            fusion.FUSED_SOFTMAX = True
    assert _rel_l2(res[True][0], res[False][0]) <= 1e-5
    assert _rel_l2(res[True][1], res[False][1]) <= 1e-4
    top = max(float(gk.norm()) for gk in res[False][2].values())
    for k, gk in res[False][2].items():
        if float(gk.norm()) > 1e-5 * top:                      # W_z.0.bias, g.bias: exactly-zero true gradients (a constant
            assert _rel_l2(res[True][2][k], gk) <= 1e-4, k     # shift in front of a train-mode BatchNorm): rounding noise only


def test_chunked_split_fp16_softmax_attention_matches_fused_kernel():
    """TPAVIModule(mode='embedded') under f16x3: the per-frame-group form on the split-fp16 contraction kernels (S = theta phi^T,
    row softmax in fp32, P g; backward with P recomputed: fusion.chunked_softmax_ok) against the fused exact-fp32 kernels, at the
    config-2 length L = 2352 (padded to 2368 as a reduction dimension), three frames in groups of two (a ragged last group):
    output within 5e-5, every gradient within 2e-4 relative L2."""
    from glfusion_amd import fusion, ops as _ops
    from glfusion_amd.models import TPAVIModule
    from oracle import glfusion_ref as orc
    res = {}
    _ops.set_precision("f16x3")
    old_bytes = fusion.CHUNK_BYTES
    try:
        for chunked in (True, False):
            fusion.CHUNKED_SOFTMAX = chunked
            fusion.CHUNK_BYTES = 2 * 2352 * 2368 * 4 + 1024         # two frames of scores per group
            m = TPAVIModule(256, mode="embedded")
            orc.closed_form_fill(m, salt=3)
            m = m.to(DEV).train()
            x = (orc.closed_form_tensor((3, 256, 3, 28, 28), 101, -1.0, 1.0) * 0.5).to(DEV).requires_grad_(True)
            assert fusion.chunked_softmax_ok(128, 2352) == chunked and fusion._frames_per_chunk(3, 2352) == 2
            z, _ = m(x)
            (z * orc.closed_form_tensor(tuple(z.shape), 102, -1.0, 1.0).to(DEV)).sum().backward()
            res[chunked] = (z.detach(), x.grad, {k: p.grad for k, p in m.named_parameters() if p.grad is not None})
    finally:
        fusion.CHUNKED_SOFTMAX, fusion.CHUNK_BYTES = True, old_bytes
        _ops.set_precision("f32")
    assert _rel_l2(res[True][0], res[False][0]) <= 5e-5
    assert _rel_l2(res[True][1], res[False][1]) <= 2e-4
    top = max(float(gk.norm()) for gk in res[False][2].values())
    for k, gk in res[False][2].items():
        if float(gk.norm()) > 1e-5 * top:
            assert _rel_l2(res[True][2][k], gk) <= 2e-4, k


def test_strided_transpose_and_padded_softmax():
    """glf_transpose2d_strided (a column slice of a wider matrix -> [cols][rows_pad], zero beyond rows) and glf_softmax_rows_ld /
    _bwd_ld (row stride > cols, padding columns written as zeros)."""
    from glfusion_amd._lib import check, lib
    b, rows, cols, ld, pad = 3, 45, 40, 100, 64
    src = rnd(b, rows, ld, seed=5).to(DEV)
    dst = torch.full((b, cols, pad), 7.0, device=DEV)
    check(lib.glf_transpose2d_strided(src[:, :, 10:].data_ptr(), ld, rows * ld, dst.data_ptr(), pad, cols * pad, rows, cols, pad, b, None), "t")
    torch.cuda.synchronize()
    assert torch.equal(dst[:, :, :rows], src[:, :, 10:10 + cols].transpose(1, 2)) and float(dst[:, :, rows:].abs().max()) == 0.0
    x = rnd(37, 64, seed=6).to(DEV)
    xs = x.clone()
    check(lib.glf_softmax_rows_ld(xs.data_ptr(), 37, 50, 64, None), "s")
    want = torch.softmax(x[:, :50].cpu(), -1)
    assert close(xs[:, :50], want, 1e-6) and float(xs[:, 50:].abs().max()) == 0.0
    dp = rnd(37, 64, seed=7).to(DEV)
    dpc = dp.clone()
    check(lib.glf_softmax_rows_bwd_ld(xs.data_ptr(), dpc.data_ptr(), 37, 50, 64, None), "sb")
    p_ = want.double()
    d_ = dp[:, :50].cpu().double()
    assert close(dpc[:, :50], p_ * (d_ - (d_ * p_).sum(-1, keepdim=True)), 1e-6) and float(dpc[:, 50:].abs().max()) == 0.0


# ------------------------------------------------------------------------------------------ C convolution entry points
@pytest.mark.parametrize("cfg", [(2, 14, 14, 64, 96, 1, 1, 0, 1), (2, 55, 55, 16, 24, 3, 2, 1, 1), (2, 28, 28, 64, 32, 3, 1, 12, 12),
                                 (2, 28, 28, 64, 32, 3, 1, 24, 24), (2, 28, 28, 64, 32, 3, 1, 36, 36), (2, 28, 28, 32, 160, 3, 1, 2, 2)])
def test_conv2d_c_entry_points_match_autograd_path(ops, cfg):
    """glf_conv2d_{fwd,dgrad,wgrad} called the way a C host would (tap-major weights, one parameter block, no policy on the
    caller's side) against ops.conv2d + autograd, which drives glf_gemm_* with the host-side policy: same results (1e-6
    relative L2: per-tap rectangle modes sum with atomics in either path)."""
    import ctypes as C
    from glfusion_amd._lib import ConvParams, ConvPlan, check, lib
    n, h, w, cin, cout, k, stride, pad, dil = cfg
    x = rnd(n, h, w, cin, seed=50).to(DEV).requires_grad_(True)
    wt = (rnd(cout, cin, k, k, seed=51) / np.sqrt(cin * k * k)).to(DEV).requires_grad_(True)
    y = ops.conv2d(x, wt, None, stride, pad, dil)
    gy = rnd(*y.shape, seed=52).to(DEV)
    y.backward(gy)
    p = ConvParams()
    p.n, p.h, p.w, p.cin, p.cout, p.kh, p.kw, p.stride, p.pad, p.dil = n, h, w, cin, cout, k, k, stride, pad, dil
    p.precision = ops.PRECISIONS.index(ops.get_precision()) + 1
    pl = ConvPlan()
    ptr = lambda t: C.c_void_p(t.data_ptr())
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    w_tap = torch.empty(k * k, cout, cin, device=DEV)
    w_tap_t = torch.empty(k * k, cin, cout, device=DEV)
    check(lib.glf_oihw_to_tap_major(ptr(wt.detach().contiguous()), ptr(w_tap), cout, cin, k * k, s), "tap_major")
    check(lib.glf_oihw_to_tap_major_t(ptr(wt.detach().contiguous()), ptr(w_tap_t), cout, cin, k * k, s), "tap_major_t")
    check(lib.glf_conv2d_plan(C.byref(p), 0, C.byref(pl)), "plan")
    y2 = torch.full((n, pl.ho, pl.wo, cout), float("nan"), device=DEV)
    check(lib.glf_conv2d_fwd(ptr(x.detach()), ptr(w_tap), None, ptr(y2), C.byref(p), s), "conv2d_fwd")
    dx2 = torch.full((n, h, w, cin), float("nan"), device=DEV)
    check(lib.glf_conv2d_dgrad(ptr(gy), ptr(w_tap), ptr(w_tap_t), ptr(dx2), C.byref(p), s), "conv2d_dgrad")
    check(lib.glf_conv2d_plan(C.byref(p), 2, C.byref(pl)), "plan")
    ws = torch.empty(max(int(pl.workspace_bytes) // 4, 1), device=DEV)
    dwt = torch.full((k * k, cout, cin), float("nan"), device=DEV)
    check(lib.glf_conv2d_wgrad(ptr(gy), ptr(x.detach()), ptr(dwt), ptr(ws), int(pl.workspace_bytes), C.byref(p), s), "conv2d_wgrad")
    dw2 = dwt.permute(1, 2, 0).reshape(cout, cin, k, k)
    assert _rel_l2(y2, y) <= 1e-6 and _rel_l2(dx2, x.grad) <= 1e-6 and _rel_l2(dw2, wt.grad) <= 1e-6
    # without a workspace the slices meet in atomics: same numbers up to summation order
    dwt2 = torch.full((k * k, cout, cin), float("nan"), device=DEV)
    check(lib.glf_conv2d_wgrad(ptr(gy), ptr(x.detach()), ptr(dwt2), None, 0, C.byref(p), s), "conv2d_wgrad(atomics)")
    assert _rel_l2(dwt2, dwt) <= 1e-6
