"""CPU: the oracle (oracle/glfusion_ref.py) against the golden vectors produced by executing
the reference's own models/ours.py (tests/golden/make_golden.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import glfusion_ref as orc

TOL = 1e-4   # north_star: masks / Dice within 1e-4 fp32


def close(a, b, tol=TOL):
    a = torch.as_tensor(a).double()
    b = torch.as_tensor(b).double()
    return bool(((a - b).abs() <= tol + tol * b.abs()).all())


@pytest.mark.parametrize("mode", ["dot", "embedded"])
def test_tpavi_unit(golden_dir, mode):
    g = np.load(os.path.join(golden_dir, f"tpavi_{mode}.npz"))
    m = orc.TPAVIModule(64, mode=mode)
    orc.closed_form_fill(m, salt=3)
    m.train()
    x = orc.closed_form_tensor((2, 64, 3, 6, 5), 101, -1.0, 1.0).requires_grad_(True)
    z, _ = m(x)
    w = orc.closed_form_tensor(tuple(z.shape), 102, -1.0, 1.0)
    (z * w).sum().backward()
    assert close(z.detach(), g["z"], 1e-5)
    assert close(x.grad, g["dx"], 1e-5)
    assert close(m.W_z[1].running_mean, g["rm"], 1e-6) and close(m.W_z[1].running_var, g["rv"], 1e-6)
    norms = dict(zip(g["grad_names"].tolist(), g["grad_norms"].tolist()))
    for name, p in m.named_parameters():
        if p.grad is None:
            assert norms[name] == -1.0
        else:
            assert abs(float(p.grad.double().norm()) - norms[name]) <= 1e-5 * max(1.0, norms[name])
    m.eval()
    with torch.no_grad():
        assert close(m(x.detach())[0], g["z_eval"], 1e-5)


def test_deeplab_head_unit(golden_dir):
    g = np.load(os.path.join(golden_dir, "deeplab_head.npz"))
    head = orc.DeepLabHead(64, 5)
    orc.closed_form_fill(head, salt=5)
    orc.set_dropout(head, 0.0)
    x = orc.closed_form_tensor((4, 64, 28, 28), 201, 0.0, 1.0).requires_grad_(True)
    head.train()
    y = head(x)
    w = orc.closed_form_tensor(tuple(y.shape), 202, -1.0, 1.0)
    (y * w).sum().backward()
    assert close(y.detach(), g["y_train"], 1e-5)
    assert close(x.grad, g["dx"], 1e-4)
    for k, v in head.state_dict().items():
        if "running" in k:
            assert close(v, g["bn:" + k], 1e-6)
    head2 = orc.DeepLabHead(64, 5)          # fresh running stats, as the generator does
    orc.closed_form_fill(head2, salt=5)
    head2.eval()
    with torch.no_grad():
        assert close(head2(x.detach()), g["y_eval"], 1e-5)


def test_state_dict_keys_match_reference(golden_dir):
    want = [ln.split(" ", 1) for ln in open(os.path.join(golden_dir, "state_dict_keys.txt")).read().splitlines()]
    m = orc.Global_and_Local(["1", "3", "4"])
    got = [(k, str(tuple(v.shape))) for k, v in m.state_dict().items()]
    assert len(got) == 1618
    assert [k for k, _ in got] == [k for k, _ in want]
    assert [s for _, s in got] == [s for _, s in want]
    assert sum(p.numel() for p in m.parameters()) == 224227623


@pytest.mark.parametrize("tag,views,n", [("c2", ["1", "3", "4"], 2), ("c1", ["1"], 8)])
def test_e2e_eval(golden_dir, tag, views, n):
    g = np.load(os.path.join(golden_dir, f"e2e_eval_{tag}.npz"))
    model = orc.Global_and_Local(views)
    orc.closed_form_fill(model, salt=1)
    model.eval()
    imgs = orc.closed_form_images(views, n)
    tgts = orc.closed_form_targets(views, n)
    with torch.no_grad():
        mask, mask_bb, fg, fl = model(imgs)
    for v in views:
        assert close(mask[v], g[f"mask:{v}"])
        assert close(mask_bb[v], g[f"mask_bb:{v}"])
        ref = torch.from_numpy(g[f"mask:{v}"])
        bits_differ = orc.binarize(mask[v]) != orc.binarize(ref)
        assert bool((ref.abs()[bits_differ] < TOL).all())      # bits may differ only where |logit| < tol
        dice = [float(x) for x in orc.overlap_metrics(tgts[v], orc.binarize(mask[v]))]
        assert np.allclose(dice, g[f"dice:{v}"], atol=TOL, rtol=0)
        for nm, f in (("fg", fg[v]), ("fl", fl[v])):
            idx = torch.from_numpy(g[f"{nm}_idx:{v}"])
            assert close(f.reshape(-1)[idx], g[f"{nm}_val:{v}"])


def test_e2e_train_step(golden_dir):
    g = np.load(os.path.join(golden_dir, "e2e_train_step.npz"))
    views, n = ["1", "3", "4"], 4
    model = orc.Global_and_Local(views)
    orc.closed_form_fill(model, salt=1)
    orc.set_dropout(model, 0.0)
    model.train()
    loss = orc.train_step(model, orc.closed_form_images(views, n), orc.closed_form_targets(views, n))
    assert abs(loss - float(g["loss"])) <= 1e-5 * abs(float(g["loss"]))
    norms = dict(zip(g["grad_names"].tolist(), g["grad_norms"].tolist()))
    norms64 = dict(zip(g["grad_names"].tolist(), g["grad_norms64"].tolist()))
    for name, p in model.named_parameters():
        if p.grad is None:
            assert norms[name] == -1.0, name
        elif name.endswith(".0.bias") and (name.startswith("init_block") or ".W_z.0." in name):
            # a conv bias feeding a train-mode BatchNorm has an exactly-zero true gradient: both sides
            # hold only rounding noise, bounded relative to the companion weight gradient
            gw = norms[name[:-4] + "weight"]
            assert float(p.grad.double().norm()) <= 1e-4 * gw and norms[name] <= 1e-4 * gw, name
        else:
            # tolerance = the larger of 2e-3 relative and 10x the reference's own fp32-vs-fp64 noise
            gn = float(p.grad.double().norm())
            tol = max(2e-3 * norms64[name], 10 * abs(norms[name] - norms64[name]), 1e-6)
            assert abs(gn - norms64[name]) <= tol, (name, gn, norms[name], norms64[name])
    sd = model.state_dict()
    for k in g.files:
        if k.startswith("bn:"):
            flat = sd[k[3:]].reshape(-1).float()
            idx = np.unique(np.linspace(0, flat.numel() - 1, num=min(9, flat.numel())).astype(np.int64))
            assert close(flat[torch.from_numpy(idx)], g[k], 1e-5), k


def test_oracle_cycle_losses_match_reference_fixture(golden_dir):
    """oracle.seg_cycle / dense_seg_cycle against vectors produced by the reference's own main.py:650-798."""
    import numpy as np
    import torch
    g = np.load(os.path.join(golden_dir, "seg_cycle.npz"))
    for tag in ("full", "small", "short"):
        T, F, scale, salt = g[f"{tag}:cfg"]
        feat0 = orc.closed_form_tensor((int(T), int(F)), int(salt), 0.0, float(scale))
        for start in (0, 5, 11):
            f = feat0.clone().requires_grad_(True)
            loss = orc.seg_cycle(f, 16, 2, 3, 10, start)
            loss.backward()
            assert abs(float(loss.detach()) - float(g[f"{tag}:seg:{start}:loss"])) <= 1e-6 * abs(float(g[f"{tag}:seg:{start}:loss"]))
            ref = torch.from_numpy(g[f"{tag}:seg:{start}:dfeat"])
            assert float((f.grad - ref).abs().max()) <= 1e-6 * float(ref.abs().max())
        for soft, overlap in ((0, 1), (1, 1), (0, 0)):
            f = feat0.clone().requires_grad_(True)
            loss = orc.dense_seg_cycle(f, 16, 2, 3, 10, bool(soft), bool(overlap))
            loss.backward()
            key = f"{tag}:dense:{soft}{overlap}"
            assert abs(float(loss.detach()) - float(g[key + ":loss"])) <= 1e-6 * abs(float(g[key + ":loss"]))
            ref = torch.from_numpy(g[key + ":dfeat"])
            assert float((f.grad - ref).abs().max()) <= 1e-6 * float(ref.abs().max())


@pytest.mark.parametrize("name", ["Global_only", "Local_only", "Global_and_Local_cyc_nofusion", "Global_and_Local_conv_merge", "Global_only_cyc_nofusion", "Foreground_and_Background",
                                  "model19", "Global_and_Local_CPS"])
def test_oracle_variants_match_reference_fixture(golden_dir, name):
    """the single-branch ablations (ours.py:1999-2249), oracle restatement vs the reference's classes."""
    g = np.load(os.path.join(golden_dir, f"variant_{name}.npz"))
    views, n = ["1", "3"], 2
    model = getattr(orc, name)(views)
    assert list(model.state_dict().keys()) == [str(k) for k in g["keys"]]
    orc.closed_form_fill(model, salt=6)
    model.eval()
    with torch.no_grad():
        out = model(orc.closed_form_images(views, n, 112, 112))
    for v in views:
        for slot, key in ((0, "mask"), (1, "mask_bb")):
            ref = torch.from_numpy(g[f"{key}:{v}"])
            flat = out[slot][v].reshape(-1)
            idx = np.unique(np.linspace(0, flat.numel() - 1, num=min(20011, flat.numel())).astype(np.int64))
            got = flat[torch.from_numpy(idx)]
            assert float((got - ref).abs().max()) <= 1e-4 * max(1.0, float(ref.abs().max())), (key, v)


def _bottleneck_pair(cls, conv, bn, inplanes, planes, stride):
    down = None
    if stride != 1 or inplanes != planes * 4:
        down = torch.nn.Sequential(conv(inplanes, planes * 4, kernel_size=1, stride=stride, bias=False), bn(planes * 4))
    return cls(inplanes, planes, stride, down)


@pytest.mark.parametrize("tag", ["plain", "down"])
def test_bottleneck_vs_reference_block(golden_dir, tag):
    """the oracle's Bottleneck against the reference's own in-tree block (models/resnet.py:43-79; fixture
    produced by executing it): train-mode forward, input / parameter gradients, running statistics, eval forward.
    What stays unpinned is torchvision's dilation rule of `_make_layer` (the in-tree block has no dilation argument)."""
    g = np.load(os.path.join(golden_dir, "bottleneck_ref.npz"))
    inplanes, planes, stride, hw = [int(v) for v in g[f"{tag}:cfg"]]
    blk = _bottleneck_pair(orc.Bottleneck, torch.nn.Conv2d, torch.nn.BatchNorm2d, inplanes, planes, stride)
    assert list(blk.state_dict().keys()) == [str(k) for k in g[f"{tag}:keys"]]
    orc.closed_form_fill(blk, salt=31)
    blk.train()
    x = orc.closed_form_tensor((4, inplanes, hw, hw), 311, -1.0, 1.0).requires_grad_(True)
    y = blk(x)
    y.backward(orc.closed_form_tensor(tuple(y.shape), 312, -1.0, 1.0))
    assert close(y.detach(), g[f"{tag}:y"], 1e-6)
    assert close(x.grad, g[f"{tag}:dx"], 1e-5)
    for k, p in blk.named_parameters():
        ref = torch.from_numpy(g[f"{tag}:g:{k}"]).double()
        assert float((p.grad.double() - ref).norm()) <= 1e-5 * float(ref.norm()) + 1e-7, k
    for k, v in blk.state_dict().items():
        if "running" in k or "num_batches" in k:
            assert close(v.float(), g[f"{tag}:bn:{k}"], 1e-6), k
    blk.eval()
    with torch.no_grad():
        assert close(blk(x.detach()), g[f"{tag}:y_eval"], 1e-6)


def kinkfree_gate(name, got, want64, scale):
    """relative L2 of a gradient against the fp64 reference; `scale` is the largest gradient norm of the parameter's
    top-level module (structurally-zero gradients -- a bias in front of a train-mode BatchNorm -- hold rounding noise only)."""
    return float((got.double() - want64.double()).norm()), 1e-3 * float(want64.double().norm()) + 1e-5 * scale


def test_kinkfree_train_step_oracle_vs_reference(golden_dir):
    """The oracle's train step on the kink-free weights (no ReLU input near zero => gradients smooth in the arithmetic)
    against the reference's fp64 evaluation of the same step: loss, logits, every parameter's gradient norm, 65-point
    gradient samples, BatchNorm running statistics."""
    g = np.load(os.path.join(golden_dir, "e2e_train_kinkfree.npz"))
    views, n = [str(v) for v in g["views"]], int(g["n"])
    model = orc.Global_and_Local(views)
    orc.kinkfree_fill(model, salt=21)
    orc.set_dropout(model, 0.0)
    model.train()
    imgs, tgts = orc.varied_images(views, n), orc.closed_form_targets(views, n)
    pred = model(imgs)[0]
    loss = sum(torch.nn.functional.binary_cross_entropy_with_logits(pred[v], tgts[v], reduction="sum") for v in views)
    loss.backward()
    assert abs(float(loss.detach()) - float(g["loss64"])) <= 1e-6 * float(g["loss64"])
    # the reference's own fp32-vs-fp64 noise on this fixture, for the record (and as a guard on the fixture's conditioning)
    n32, n64 = g["grad_norms32"], g["grad_norms64"]
    live = n64 > 1e-6 * n64.max()
    assert np.median(np.abs(n32[live] - n64[live]) / n64[live]) < 1e-5
    for v in views:
        flat = pred[v].detach().reshape(-1)
        idx = np.unique(np.linspace(0, flat.numel() - 1, num=20011).astype(np.int64))
        assert close(flat[torch.from_numpy(idx)], g[f"mask64:{v}"], 1e-4), v
    names = [str(k) for k in g["grad_names"]]
    norms64 = dict(zip(names, n64.tolist()))
    noise32 = dict(zip(names, g["grad_noise32"].tolist()))
    assert float(np.median(g["grad_noise32"][live])) < 1e-4        # the reference's fp32-vs-fp64 relative L2, median over tensors
    scale = {}
    for k, v in norms64.items():
        scale[k.split(".")[0]] = max(scale.get(k.split(".")[0], 0.0), v)
    for name, p in model.named_parameters():
        if norms64[name] < 0:
            assert p.grad is None, name
            continue
        gn = float(p.grad.double().norm())
        tol = max(1e-3, 5.0 * noise32[name]) * norms64[name] + 1e-5 * scale[name.split(".")[0]]
        assert abs(gn - norms64[name]) <= tol, (name, gn, norms64[name])
        flat = p.grad.reshape(-1)
        idx = np.unique(np.linspace(0, flat.numel() - 1, num=min(65, flat.numel())).astype(np.int64))
        want = torch.from_numpy(g["g64:" + name]).double()
        err = float((flat[torch.from_numpy(idx)].double() - want).norm())
        assert err <= 1e-3 * float(want.norm()) + tol * (len(idx) / flat.numel()) ** 0.5 + 1e-12, (name, err, float(want.norm()))
    sd = model.state_dict()
    for k in g.files:
        if k.startswith("bn:"):
            flat = sd[k[3:]].reshape(-1).float()
            idx = np.unique(np.linspace(0, flat.numel() - 1, num=min(9, flat.numel())).astype(np.int64))
            assert close(flat[torch.from_numpy(idx)], g[k], 1e-5), k
