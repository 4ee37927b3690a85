#!/usr/bin/env python3
"""Generate tests/golden/*.npz by EXECUTING THE REFERENCE's own models/ours.py on CPU.

Runs only in the build container (needs /root/reference); the fixtures it writes are
plain data (inputs are closed-form, outputs are arrays) and travel with the repo.

Recipe: the reference imports three packages that are absent
here and unused by the model code -- monai.data.DataLoader, tensorboardX.SummaryWriter --
plus torchvision.models.resnet (absent; supplies ResNet-50).  We register in-memory
stand-ins before importing: the first two are empty placeholders, the third exposes the
oracle's own ResNet-50 restatement (oracle/glfusion_ref.py::resnet50).  Everything else on
the path (conv1 swap, DeepLabHead/ASPP, TPAVIModule, Global_and_Local wiring) is the
reference's code, executed verbatim.

Weights are never stored: both sides fill state_dict() with oracle.closed_form_fill.
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/GLfusion"
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)

from oracle import glfusion_ref as orc  # noqa: E402


def import_reference():
    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    monai = stub("monai")
    monai.data = stub("monai.data", DataLoader=object)
    stub("tensorboardX", SummaryWriter=object)
    tv = stub("torchvision")
    tv.models = stub("torchvision.models")
    tv.models.resnet = stub("torchvision.models.resnet", resnet50=orc.resnet50)
    sys.path.insert(0, REF)
    import models.ours as ours          # noqa: E402
    import models.deeplabv3 as dl       # noqa: E402
    return ours, dl


def sample_idx(n: int, k: int = 257) -> np.ndarray:
    return np.unique(np.linspace(0, n - 1, num=min(k, n)).astype(np.int64))


def t2n(t: torch.Tensor) -> np.ndarray:
    return t.detach().cpu().contiguous().numpy().copy()       # a copy: never an alias of module state


def grads_summary(model: torch.nn.Module):
    """per-parameter L2 norm and a strided sample of each gradient (None -> norm -1)."""
    names, norms, samples = [], [], {}
    for name, p in model.named_parameters():
        names.append(name)
        if p.grad is None:
            norms.append(-1.0)
            continue
        g = p.grad.detach().double().reshape(-1)
        norms.append(float(g.norm()))
        samples[name] = t2n(p.grad.reshape(-1)[torch.from_numpy(sample_idx(g.numel(), 33))])
    return names, np.array(norms), samples


def main() -> None:
    only = set(sys.argv[1:])          # optional: tpavi head eval train cycle
    want = lambda name: not only or name in only
    torch.manual_seed(0)
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    ours, dl = import_reference()
    out = {}

    # ------------------------------------------------------------------ unit: TPAVIModule
    for mode in (("dot", "embedded") if want("tpavi") else ()):
        m = ours.TPAVIModule(in_channels=64, mode=mode)
        orc.closed_form_fill(m, salt=3)
        m.train()
        x = orc.closed_form_tensor((2, 64, 3, 6, 5), 101, -1.0, 1.0).requires_grad_(True)
        z, _ = m(x)
        w = orc.closed_form_tensor(tuple(z.shape), 102, -1.0, 1.0)
        (z * w).sum().backward()
        names, norms, samples = grads_summary(m)
        d = {"z": t2n(z), "dx": t2n(x.grad), "grad_names": np.array(names), "grad_norms": norms,
             "rm": t2n(m.W_z[1].running_mean), "rv": t2n(m.W_z[1].running_var)}
        d.update({"g:" + k: v for k, v in samples.items()})
        m.eval()
        with torch.no_grad():
            d["z_eval"] = t2n(m(x.detach())[0])
        np.savez_compressed(os.path.join(HERE, f"tpavi_{mode}.npz"), **d)
        print("tpavi", mode, "|z|max", float(z.abs().max()))

    # ------------------------------------------------------------------ unit: DeepLabHead
    if not want("head"):
        head = None
    head = dl.DeepLabHead(64, 5)
    orc.closed_form_fill(head, salt=5)
    orc.set_dropout(head, 0.0)
    x = orc.closed_form_tensor((4, 64, 28, 28), 201, 0.0, 1.0).requires_grad_(True)   # N=4: the pooled-branch BN needs > 2 samples
    head.train()
    y = head(x)
    w = orc.closed_form_tensor(tuple(y.shape), 202, -1.0, 1.0)
    (y * w).sum().backward()
    names, norms, samples = grads_summary(head)
    d = {"y_train": t2n(y), "dx": t2n(x.grad), "grad_names": np.array(names), "grad_norms": norms}
    d.update({"g:" + k: v for k, v in samples.items()})
    d.update({"bn:" + k: t2n(v) for k, v in head.state_dict().items() if "running" in k})
    head2 = dl.DeepLabHead(64, 5)
    orc.closed_form_fill(head2, salt=5)
    head2.eval()
    with torch.no_grad():
        d["y_eval"] = t2n(head2(x.detach()))
    np.savez_compressed(os.path.join(HERE, "deeplab_head.npz"), **d)
    print("head |y|max", float(y.abs().max()))

    # ------------------------------------------------------------------ e2e eval, config-2 views, N=2
    for tag, views, n in ((("c2", ["1", "3", "4"], 2), ("c1", ["1"], 8)) if want("eval") else ()):
        model = ours.Global_and_Local(view_num=views)
        keys = list(model.state_dict().keys())
        orc.closed_form_fill(model, salt=1)
        model.eval()
        imgs = orc.closed_form_images(views, n)
        tgts = orc.closed_form_targets(views, n)
        with torch.no_grad():
            mask, mask_bb, fg, fl = model(imgs)
        d = {"n_keys": np.array(len(keys)), "keys_head": np.array(keys[:8] + keys[-8:])}
        for v in views:
            d[f"mask:{v}"] = t2n(mask[v])
            d[f"mask_bb:{v}"] = t2n(mask_bb[v])
            pred = torch.where(torch.sigmoid(mask[v]) > 0.5, 1, 0)
            # reference metric code path: main.py:800-815 restated in the oracle; the dice of the
            # reference logits vs the closed-form targets
            d[f"dice:{v}"] = np.array([float(x) for x in orc.overlap_metrics(tgts[v], pred)])
            d[f"pos_frac:{v}"] = np.array(float(pred.float().mean()))
            d[f"near_zero:{v}"] = np.array(int((mask[v].abs() < 1e-4).sum()))
            for nm, f in (("fg", fg[v]), ("fl", fl[v])):
                flat = f.reshape(-1)
                idx = sample_idx(flat.numel(), 4099)
                d[f"{nm}_idx:{v}"] = idx
                d[f"{nm}_val:{v}"] = t2n(flat[torch.from_numpy(idx)])
                d[f"{nm}_sum:{v}"] = np.array([float(f.double().sum()), float(f.double().abs().sum())])
            print(tag, v, "pos_frac", float(d[f"pos_frac:{v}"]), "dice", d[f"dice:{v}"][1],
                  "near0", int(d[f"near_zero:{v}"]), "|mask|max", float(mask[v].abs().max()))
        np.savez_compressed(os.path.join(HERE, f"e2e_eval_{tag}.npz"), **d)
        if tag == "c2":
            with open(os.path.join(HERE, "state_dict_keys.txt"), "w") as fh:
                for k in keys:
                    fh.write(f"{k} {tuple(model.state_dict()[k].shape)}\n")

    # ------------------------------------------------------------------ train-mode step, dropout p = 0
    if not want("train"):
        return
    views, n = ["1", "3", "4"], 4          # N=4: train-mode BN over 2 pooled samples is degenerate (output +-1, zero gradient)
    model = ours.Global_and_Local(view_num=views)
    orc.closed_form_fill(model, salt=1)
    orc.set_dropout(model, 0.0)
    model.train()
    imgs = orc.closed_form_images(views, n)
    tgts = orc.closed_form_targets(views, n)
    pred, _, _, _ = model(imgs)
    bce = torch.nn.BCEWithLogitsLoss(reduction="sum")          # main.py:87
    loss = sum(bce(pred[v], tgts[v]) for v in views)           # main.py:209-211
    loss.backward()                                            # main.py:241
    names, norms, samples = grads_summary(model)
    d = {"loss": np.array(float(loss)), "grad_names": np.array(names), "grad_norms": norms}
    # the same step executed by the reference in float64: lets the tests size each gradient's
    # tolerance by the reference's own fp32 rounding noise (BN over N=2 frames is ill-conditioned)
    m64 = ours.Global_and_Local(view_num=views)
    orc.closed_form_fill(m64, salt=1)
    orc.set_dropout(m64, 0.0)
    m64.double().train()
    p64, _, _, _ = m64({v: imgs[v].double() for v in views})
    l64 = sum(bce(p64[v], tgts[v].double()) for v in views)
    l64.backward()
    _, norms64, samples64 = grads_summary(m64)
    d["loss64"] = np.array(float(l64))
    for v in views:
        d[f"mask64:{v}"] = t2n(p64[v].float())
    d["grad_norms64"] = norms64
    d.update({"g64:" + k: v for k, v in samples64.items()})
    del m64, p64, l64
    d.update({"g:" + k: v for k, v in samples.items()})
    for v in views:
        d[f"mask:{v}"] = t2n(pred[v])
    sd = model.state_dict()
    for k in sd:
        if ("running_mean" in k or "running_var" in k or "num_batches" in k) and not k.startswith("network."):
            # store a strided sample of every running stat (3 momentum updates for classifier BNs)
            flat = sd[k].reshape(-1).float()
            d["bn:" + k] = t2n(flat[torch.from_numpy(sample_idx(flat.numel(), 9))])
    np.savez_compressed(os.path.join(HERE, "e2e_train_step.npz"), **d)
    print("train loss", float(loss))


def bottleneck_fixture() -> None:
    """the reference's OWN in-tree Bottleneck (models/resnet.py:43-79 -- the block torchvision's
    ResNet-50 is made of, stride on the 3x3, no dilation argument), train-mode forward / backward / running statistics,
    for the identity-shortcut form and the stride-2 + downsample form (models/resnet.py:110-117)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_resnet", os.path.join(REF, "models", "resnet.py"))
    ref_resnet = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref_resnet)
    d = {}
    for tag, (inplanes, planes, stride, hw) in {"plain": (64, 16, 1, 14), "down": (64, 32, 2, 15)}.items():
        for dt, sfx in ((torch.float32, ""), (torch.float64, "64")):
            down = None
            if stride != 1 or inplanes != planes * 4:           # models/resnet.py:110-117
                down = torch.nn.Sequential(torch.nn.Conv2d(inplanes, planes * 4, kernel_size=1, stride=stride, bias=False),
                                           torch.nn.BatchNorm2d(planes * 4))
            blk = ref_resnet.Bottleneck(inplanes, planes, stride, down)
            orc.closed_form_fill(blk, salt=31)
            blk = blk.to(dt).train()
            x = orc.closed_form_tensor((4, inplanes, hw, hw), 311, -1.0, 1.0).to(dt).requires_grad_(True)
            y = blk(x)
            gy = orc.closed_form_tensor(tuple(y.shape), 312, -1.0, 1.0).to(dt)
            y.backward(gy)
            d[f"{tag}:y{sfx}"] = t2n(y.float())
            d[f"{tag}:dx{sfx}"] = t2n(x.grad.float())
            for k, p_ in blk.named_parameters():
                d[f"{tag}:g{sfx}:{k}"] = t2n(p_.grad.float())
            if not sfx:
                d[f"{tag}:cfg"] = np.array([inplanes, planes, stride, hw])
                d[f"{tag}:keys"] = np.array(list(blk.state_dict().keys()))
                for k, v in blk.state_dict().items():
                    if "running" in k or "num_batches" in k:
                        d[f"{tag}:bn:{k}"] = t2n(v.float())
                blk.eval()
                with torch.no_grad():
                    d[f"{tag}:y_eval"] = t2n(blk(x.detach()))
        print("bottleneck", tag, "|y|max", float(np.abs(d[f"{tag}:y"]).max()))
    np.savez_compressed(os.path.join(HERE, "bottleneck_ref.npz"), **d)


def kinkfree_fixture(ours) -> None:
    """A train step whose gradients are smooth in the arithmetic (oracle.kinkfree_fill: no ReLU input near zero), executed
    by the reference in fp32 and fp64: per-parameter gradient norms, 65-point gradient samples, logits samples."""
    views, n = ["1", "3"], 8
    imgs = orc.varied_images(views, n)
    tgts = orc.closed_form_targets(views, n)
    bce = torch.nn.BCEWithLogitsLoss(reduction="sum")
    d = {"views": np.array(views), "n": np.array(n)}
    g32 = {}
    for dt, sfx in ((torch.float32, "32"), (torch.float64, "64")):
        model = ours.Global_and_Local(view_num=views)
        orc.kinkfree_fill(model, salt=21)
        orc.set_dropout(model, 0.0)
        model = model.to(dt).train()
        pred = model({v: imgs[v].to(dt) for v in views})[0]
        loss = sum(bce(pred[v], tgts[v].to(dt)) for v in views)
        loss.backward()
        d["loss" + sfx] = np.array(float(loss.detach()))
        names, norms, noise = [], [], []
        for name, p_ in model.named_parameters():
            names.append(name)
            if p_.grad is None:
                norms.append(-1.0)
                noise.append(-1.0)
                continue
            g = p_.grad.detach().double().reshape(-1)
            norms.append(float(g.norm()))
            if sfx == "32":
                g32[name] = p_.grad.detach().reshape(-1).clone()
            else:
                d["g64:" + name] = t2n(g[torch.from_numpy(sample_idx(g.numel(), 65))].float())
                # the reference's OWN fp32-vs-fp64 deviation on this tensor (relative L2): the yardstick for any fp32 engine
                noise.append(float((g32.pop(name).double() - g).norm()) / max(float(g.norm()), 1e-300))
        if sfx == "64":
            d["grad_noise32"] = np.array(noise)
        d["grad_names"] = np.array(names)
        d["grad_norms" + sfx] = np.array(norms)
        for v in views:
            flat = pred[v].detach().reshape(-1)
            d[f"mask{sfx}:{v}"] = t2n(flat[torch.from_numpy(sample_idx(flat.numel(), 20011))].float())
        if sfx == "32":
            sd = model.state_dict()
            for k in sd:
                if ("running_mean" in k or "running_var" in k) and not k.startswith("network."):
                    flat = sd[k].reshape(-1).float()
                    d["bn:" + k] = t2n(flat[torch.from_numpy(sample_idx(flat.numel(), 9))])
        print("kinkfree", sfx, "loss", float(loss.detach()))
        del model, pred, loss
    np.savez_compressed(os.path.join(HERE, "e2e_train_kinkfree.npz"), **d)


def variants_fixture(ours) -> None:
    """the two single-branch ablations, eval() outputs of the reference's own classes."""
    views, n = ["1", "3"], 2
    imgs = orc.closed_form_images(views, n, 112, 112)
    only = [a for a in sys.argv[1:] if a not in ("variants",)]
    for name in ("Global_only", "Local_only", "Global_and_Local_cyc_nofusion", "Global_and_Local_conv_merge", "Global_only_cyc_nofusion", "Foreground_and_Background",
                 "model19", "Global_and_Local_CPS"):
        if only and name not in only:
            continue
        model = getattr(ours, name)(views)
        orc.closed_form_fill(model, salt=6)
        model.eval()
        with torch.no_grad():
            out = model(imgs)
        d = {"keys": np.array(list(model.state_dict().keys()))}
        for v in views:
            # a 20 011-point strided sample of each logit map (full maps for every variant would be 13 MB of fixtures)
            idx = torch.from_numpy(sample_idx(out[0][v].numel(), 20011))
            d[f"mask:{v}"] = t2n(out[0][v].reshape(-1)[idx])
            d[f"mask_bb:{v}"] = t2n(out[1][v].reshape(-1)[idx])
            third = out[2][v]
            d[f"third:{v}"] = t2n(third) if third.shape[1] == 1 else t2n(third.reshape(-1)[torch.from_numpy(sample_idx(third.numel(), 4099))])
            if out[3] is not None:
                fo = out[3][v]
                d[f"fourth:{v}"] = t2n(fo.reshape(-1)[torch.from_numpy(sample_idx(fo.numel(), 4099))])
        np.savez_compressed(os.path.join(HERE, f"variant_{name}.npz"), **d)
        print(name, "done", {v: float(out[0][v].abs().mean()) for v in views})


def reference_cycle_functions():
    """Trainer.seg_cycle / Trainer.dense_seg_cycle (main.py:650-798) as callables.  main.py cannot be imported (it
    needs utils.PCGrad and the authors' data files), so the two method definitions are cut out of its syntax tree
    and compiled on their own -- the reference's code, executed verbatim, nothing of it is written to disk.
    `np.random.choice` (the random start frame, main.py:655) is replaced by a stub that returns `forced_start`."""
    import ast
    tree = ast.parse(open(os.path.join(REF, "main.py")).read())
    fns = [n for cls in tree.body if isinstance(cls, ast.ClassDef) and cls.name == "Trainer"
           for n in cls.body if isinstance(n, ast.FunctionDef) and n.name in ("seg_cycle", "dense_seg_cycle")]
    assert len(fns) == 2
    mod = ast.Module(body=fns, type_ignores=[])
    state = {"start": 0}
    fake_np = types.SimpleNamespace(random=types.SimpleNamespace(choice=lambda n: state["start"]))
    ns = {"torch": torch, "np": fake_np}
    exec(compile(mod, "<reference main.py:650-798>", "exec"), ns)
    return ns["seg_cycle"], ns["dense_seg_cycle"], state


def cycle_fixture() -> None:
    seg_cycle, dense_seg_cycle, state = reference_cycle_functions()
    me = types.SimpleNamespace(device=torch.device("cpu"))
    d = {}
    # the call of main.py:227-231: T = 40 frames, target_region 16, cyc_off 2, chunk_size 3, temperature 10
    for tag, (T, F, scale) in {"full": (40, 2048, 30.0), "small": (40, 64, 3.0), "short": (29, 40, 2.0)}.items():
        feat0 = orc.closed_form_tensor((T, F), 700 + F, 0.0, scale)
        d[f"{tag}:cfg"] = np.array([T, F, scale, 700 + F], dtype=np.float64)
        for start in (0, 5, 11):
            state["start"] = start
            feat = feat0.clone().requires_grad_(True)
            loss = seg_cycle(me, feat, target_region=16, cyc_off=2, chunk_size=3, temperature=10)
            loss.backward()
            d[f"{tag}:seg:{start}:loss"] = t2n(loss)
            d[f"{tag}:seg:{start}:dfeat"] = t2n(feat.grad)
        for soft, overlap in ((False, True), (True, True), (False, False)):
            feat = feat0.clone().requires_grad_(True)
            loss = dense_seg_cycle(me, feat, target_region=16, cyc_off=2, chunk_size=3, temperature=10,
                                   soft_label=soft, is_overlap=overlap)
            loss.backward()
            d[f"{tag}:dense:{int(soft)}{int(overlap)}:loss"] = t2n(loss)
            d[f"{tag}:dense:{int(soft)}{int(overlap)}:dfeat"] = t2n(feat.grad)
    np.savez_compressed(os.path.join(HERE, "seg_cycle.npz"), **d)
    print("cycle fixture:", {k: float(v) for k, v in d.items() if k.endswith("loss")})


if __name__ == "__main__":
    args = sys.argv[1:]
    if not args or "cycle" in args:
        cycle_fixture()
    if not args or "bottleneck" in args:
        bottleneck_fixture()
    if not args or "kinkfree" in args:
        torch.set_num_threads(max(1, os.cpu_count() or 1))
        kinkfree_fixture(import_reference()[0])
    if not args or "variants" in args:
        torch.set_num_threads(max(1, os.cpu_count() or 1))
        variants_fixture(import_reference()[0])
    if not args or set(args) & {"tpavi", "head", "eval", "train"}:
        main()
