"""End-to-end probe of the 16-bit-storage mode ("bf16" precision): eval logits / Dice against the reference fixture, the
kink-free train step against the fp64 oracle (per-tensor gradient deviations), and a C2-sized step's time and peak memory.
Run on the GPU box: python profiles/ubench/s16_e2e_probe.py [--time]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from glfusion_amd import ops  # noqa: E402
from glfusion_amd.models import Global_and_Local  # noqa: E402
from oracle import glfusion_ref as orc  # noqa: E402

DEV = "cuda"
GOLD = os.path.join(ROOT, "tests", "golden")


def eval_parity(prec):
    ops.set_precision(prec)
    g = np.load(os.path.join(GOLD, "e2e_eval_c2.npz"))
    views, n = ["1", "3", "4"], 2
    model = Global_and_Local(views)
    orc.closed_form_fill(model, salt=1)
    model = model.to(DEV).eval()
    imgs = {v: t.to(DEV) for v, t in orc.closed_form_images(views, n).items()}
    tgts = orc.closed_form_targets(views, n)
    with torch.no_grad():
        mask = model(imgs)[0]
    for v in views:
        ref = torch.from_numpy(g[f"mask:{v}"])
        got = mask[v].float().cpu()
        err = float((got - ref).abs().max()) / float(ref.abs().max())
        rl2 = float((got - ref).norm() / ref.norm())
        dice = ops.overlap_metrics_from_counts(ops.overlap_counts(mask[v], tgts[v].to(DEV)))
        flips = int(((got > 0) != (ref > 0)).sum())
        print(f"[{prec}] eval view {v}: max err / max|ref| {err:.3e}  rel-L2 {rl2:.3e}  dice {dice[1]:.5f} vs {float(g[f'dice:{v}'][1]):.5f}  "
              f"mask flips {flips}/{ref.numel()}")


def train_parity(prec, salt=21):
    ops.set_precision(prec)
    g = np.load(os.path.join(GOLD, "e2e_train_kinkfree.npz"))
    views, n = [str(v) for v in g["views"]], int(g["n"])
    ref = orc.Global_and_Local(views)
    orc.kinkfree_fill(ref, salt=salt)
    orc.set_dropout(ref, 0.0)
    ref = ref.double().train()
    imgs, tgts = orc.varied_images(views, n), orc.closed_form_targets(views, n)
    pred = ref({v: imgs[v].double() for v in views})[0]
    loss_ref = sum(torch.nn.functional.binary_cross_entropy_with_logits(pred[v], tgts[v].double(), reduction="sum") for v in views)
    loss_ref.backward()
    want = {k: p.grad.clone() for k, p in ref.named_parameters() if p.grad is not None}
    model = Global_and_Local(views)
    orc.kinkfree_fill(model, salt=salt)
    orc.set_dropout(model, 0.0)
    model = model.to(DEV).train()
    out = model({v: imgs[v].to(DEV) for v in views})[0]
    loss = sum(ops.bce_with_logits_sum(out[v], tgts[v].to(DEV)) for v in views)
    loss.backward()
    torch.cuda.synchronize()
    print(f"[{prec}] train: loss {float(loss):.4f} vs {float(loss_ref):.4f} rel {abs(float(loss) - float(loss_ref)) / float(loss_ref):.2e}")
    for v in views:
        e = float((out[v].detach().float().cpu().double() - pred[v].detach()).norm() / pred[v].detach().norm())
        print(f"   logits view {v} rel-L2 {e:.3e}")
    by = {}
    for name, p in model.named_parameters():
        if name not in want:
            continue
        w = want[name]
        if float(w.norm()) <= 1e-2 * max(float(x.norm()) for kk, x in want.items() if kk.split(".")[0] == name.split(".")[0]):
            continue
        got = p.grad.detach().cpu().double()
        rel = float((got - w).norm() / w.norm())
        cos = float((got * w).sum() / (got.norm() * w.norm()).clamp_min(1e-300))
        by.setdefault(name.split(".")[0], []).append((rel, cos, name))
    for k, lst in by.items():
        lst.sort(reverse=True)
        rels = np.array([r for r, _, _ in lst])
        print(f"   {k:12s} n={len(lst):3d} rel-L2 median {np.median(rels):.3e} p90 {np.percentile(rels, 90):.3e} max {rels.max():.3e} ({lst[0][2]}, cos {lst[0][1]:.4f})"
              f"  min cos {min(c for _, c, _ in lst):.4f}")


def natural_parity(prec, views=("1", "3"), n=4, offset=0.0):
    """default (kaiming) initialisation, W_z BatchNorm gamma ~ N(1, 0.1): the bench's weights; truth = the oracle in float64.
    offset > 0: every BatchNorm2d beta at +-offset by channel parity (kink-free, but with ZERO-MEAN conv weights)."""
    ops.set_precision(prec)
    views = list(views)
    torch.manual_seed(0)
    model = Global_and_Local(views)
    with torch.no_grad():
        for m in (model.global_attn, model.local_attn):
            m.W_z[1].weight.normal_(1.0, 0.1)
        if offset > 0:
            for m in model.modules():
                if isinstance(m, torch.nn.BatchNorm2d):
                    c = m.num_features
                    m.bias.copy_(torch.where(torch.arange(c) % 2 == 0, 1.0, -1.0) * offset)
                    m.weight.uniform_(0.9, 1.1)
                elif isinstance(m, torch.nn.Conv2d) and m.out_channels == 1 and m.bias is not None:
                    m.weight.mul_(0.02); m.bias.fill_(-3.0)
    orc.set_dropout(model, 0.0)
    ref = orc.Global_and_Local(views)
    ref.load_state_dict(model.state_dict(), strict=True)
    orc.set_dropout(ref, 0.0)
    ref = ref.double().train()
    imgs, tgts = orc.varied_images(views, n), orc.closed_form_targets(views, n)
    pred = ref({v: imgs[v].double() for v in views})[0]
    loss_ref = sum(torch.nn.functional.binary_cross_entropy_with_logits(pred[v], tgts[v].double(), reduction="sum") for v in views)
    loss_ref.backward()
    want = {k: p.grad.clone() for k, p in ref.named_parameters() if p.grad is not None}
    model = model.to(DEV).train()
    out = model({v: imgs[v].to(DEV) for v in views})[0]
    loss = sum(ops.bce_with_logits_sum(out[v], tgts[v].to(DEV)) for v in views)
    loss.backward()
    torch.cuda.synchronize()
    print(f"[{prec}] natural: loss {float(loss.detach()):.4f} vs {float(loss_ref):.4f} rel {abs(float(loss.detach()) - float(loss_ref)) / float(loss_ref):.2e}")
    scale = {}
    for k, w in want.items():
        scale[k.split(".")[0]] = max(scale.get(k.split(".")[0], 0.0), float(w.norm()))
    by = {}
    for name, p in model.named_parameters():
        if name not in want:
            continue
        w = want[name]
        if float(w.norm()) <= 1e-2 * scale[name.split(".")[0]]:
            continue
        got = p.grad.detach().cpu().double()
        rel = float((got - w).norm() / w.norm())
        cos = float((got * w).sum() / (got.norm() * w.norm()).clamp_min(1e-300))
        by.setdefault(name.split(".")[0], []).append((rel, cos, name))
    for k, lst in by.items():
        lst.sort(reverse=True)
        rels = np.array([r for r, _, _ in lst])
        print(f"   {k:12s} n={len(lst):3d} rel-L2 median {np.median(rels):.3e} p90 {np.percentile(rels, 90):.3e} max {rels.max():.3e} ({lst[0][2]}, cos {lst[0][1]:.4f})"
              f"  min cos {min(c for _, c, _ in lst):.4f}")


def time_step(prec, steps=6):
    ops.set_precision(prec)
    views = ["1", "3", "4"]
    torch.manual_seed(0)
    model = Global_and_Local(views).to(DEV).train()
    with torch.no_grad():
        for m in (model.global_attn, model.local_attn):
            m.W_z[1].weight.normal_(1.0, 0.1)
    g = torch.Generator(device=DEV).manual_seed(1234)
    n = 64
    imgs = {v: torch.rand(n, 1, 112, 112, device=DEV, generator=g) for v in views}
    tgts = {v: (torch.rand(n, 5, 112, 112, device=DEV, generator=g) < 0.3).float() for v in views}
    ts = []
    for i in range(steps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for p in model.parameters():
            p.grad = None
        out = model(imgs)[0]
        loss = sum(ops.bce_with_logits_sum(out[v], tgts[v]) for v in views)
        loss.backward()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    print(f"[{prec}] C2 step ms: {['%.1f' % t for t in ts]}  loss {float(loss):.1f}  peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")


if __name__ == "__main__":
    torch.set_num_threads(16)
    if "--natural" in sys.argv:
        for prec in ("bf16", "f16", "f16x3"):
            natural_parity(prec)
        sys.exit(0)
    if "--zm" in sys.argv:
        for off in (3.0, 6.0):
            for prec in ("bf16", "f16", "f16x3"):
                print("offset", off)
                natural_parity(prec, offset=off)
        sys.exit(0)
    eval_parity("bf16")
    eval_parity("f16")
    train_parity("bf16")
    train_parity("f16")
    if "--time" in sys.argv:
        time_step("bf16")
        torch.cuda.empty_cache(); torch.cuda.reset_peak_memory_stats()
        time_step("f16x3")
