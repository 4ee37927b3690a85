"""Caller-side rows f1 / f2 end to end: the reference's inner training step (main.py:202-243) with the cycle term and
the Adam update, HIP engine vs the oracle on the CPU."""
import os

import numpy as np
import pytest
import torch

from oracle import glfusion_ref as orc   # the checker (tests only)

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_training_step_with_cycle_loss_and_adam_vs_oracle():
    from glfusion_amd import ops
    from glfusion_amd.models import Global_and_Local
    from glfusion_amd.optim import Adam
    views, n, t_video, start, temp = ["1"], 4, 29, 5, 0.02
    ref = orc.Global_and_Local(views)
    orc.closed_form_fill(ref, salt=5)
    orc.set_dropout(ref, 0.0)
    model = Global_and_Local(views)
    model.load_state_dict(ref.state_dict(), strict=True)
    orc.set_dropout(model, 0.0)
    model = model.to(DEV).train()
    ref.train()
    imgs = orc.closed_form_images(views, n, 112, 112)
    tgts = orc.closed_form_targets(views, n)
    video = {v: orc.closed_form_tensor((t_video, 1, 112, 112), 77, 0.0, 1.0) for v in views}
    # (the closed-form weights give pooled features of ~1e3; with the shipped temperature 10 the chunk soft-max is
    # saturated and the loss's own fp32 conditioning, measured against fp64, is 1e-4 for BOTH sides -- the kernel is
    # pinned on that regime in test_gpu_cycle.py; here the temperature keeps the plumbing test well-conditioned)

    # oracle: main.py:207-241 (seg + 1e-2 * cyc), then torch.optim.Adam (main.py:162-165)
    params_ref = [p for nm, p in ref.named_parameters() if not nm.startswith("network.")]
    opt_ref = torch.optim.Adam(params_ref, lr=3e-4, weight_decay=1e-5)
    pred = ref(imgs)[0]
    seg = sum(torch.nn.functional.binary_cross_entropy_with_logits(pred[v], tgts[v], reduction="sum") for v in views)
    feat = ref(video)[2]
    cyc = sum(orc.seg_cycle(feat[v].sum(dim=(2, 3)), 16, 2, 3, temp, start) for v in views)
    total_ref = seg + 1e-2 * cyc
    opt_ref.zero_grad()
    total_ref.backward()
    gref = {nm: p.grad.clone() for nm, p in ref.named_parameters() if p.grad is not None}
    opt_ref.step()

    # engine
    opt = Adam([p for nm, p in model.named_parameters() if not nm.startswith("network.")], lr=3e-4, weight_decay=1e-5)
    pred = model({v: t.to(DEV) for v, t in imgs.items()})[0]
    seg = sum(ops.bce_with_logits_sum(pred[v], tgts[v].to(DEV)) for v in views)
    feats = ops.pooled_fusion_features(model({v: t.to(DEV) for v, t in video.items()})[2])
    cyc_e = sum(ops.seg_cycle(feats[v], 16, 2, 3, temp, start) for v in views)
    total = seg + 1e-2 * cyc_e
    opt.zero_grad()
    total.backward()

    assert abs(float(cyc_e) - float(cyc)) <= 1e-3 * abs(float(cyc)) + 1e-6          # through a 53-layer fp32 network
    assert abs(float(total) - float(total_ref)) <= 2e-5 * abs(float(total_ref))
    # gradients that the cycle term reaches only through the global fusion block
    top = max(float(w.norm()) for nm, w in gref.items() if nm.startswith("global_attn."))
    for nm, p in model.named_parameters():
        if nm.startswith("global_attn.") and nm in gref and not nm.startswith("global_attn.align_channel"):
            g, w = p.grad.double().cpu(), gref[nm].double()
            # fp32 gradients of this network carry ~0.3-0.5 % of rounding noise between ANY two fp32 evaluations (the
            # reference against its own fp64 run: tests/golden/e2e_train_step.npz, grad_norms vs grad_norms64), and
            # parameters whose true gradient is zero (a bias in front of a train-mode BatchNorm) hold nothing else
            assert float((g - w).norm()) <= 2e-2 * float(w.norm()) + 1e-3 * top, nm
    opt.step()
    # The first Adam step moves every element by lr * g / (|g| + eps) ~ lr * sign(g): elements whose gradient is
    # pure rounding noise (true value 0) may legitimately move the other way, everything else must land together.
    n_el, n_off, moved = 0, 0, 0
    want = dict(ref.named_parameters())
    for nm, p in model.named_parameters():
        if nm.startswith("network.") or nm not in gref:
            continue
        a, b = p.detach().double().cpu(), want[nm].detach().double()
        n_el += a.numel()
        n_off += int(((a - b).abs() > 0.1 * 3e-4).sum())
        moved += 1
    assert moved > 100
    assert n_off <= 0.01 * n_el, (n_off, n_el)


@pytest.mark.parametrize("precision", [None, "bf16"])
def test_trainer_surface_runs_cycle_training_and_checkpoints(tmp_path, precision):
    """precision "bf16": the same Trainer surface in 16-bit storage mode (config['train']['precision']) -- seg + cycle loss, fused
    Adam on fp32 master weights, reference-format checkpoint, eval harness."""
    from glfusion_amd import ops
    from glfusion_amd.engine import Trainer
    cfg = {"train": {"batch_size": 2, "num_epochs": 1, "clip_length": 29, "view_num": ["1"], "test_view": ["1"], "dense_cyc": True,
                     "save_dir": str(tmp_path), "iters_per_epoch": 1, "global_rank": 0},
           "net": {"opt": {"opt_name": "Adam", "lr": 3e-4, "params": (0.9, 0.999), "weight_decay": 1e-5}}}
    if precision:
        cfg["train"]["precision"] = precision
    try:
        _trainer_surface(cfg, tmp_path)
    finally:
        ops.set_precision("f32")


def _trainer_surface(cfg, tmp_path):
    from glfusion_amd.engine import Trainer
    t = Trainer(cfg)
    before = t.model.classifier["1"][4].weight.detach().clone()
    t.train(is_backbone=False, is_cycle=True)
    assert not torch.equal(before, t.model.classifier["1"][4].weight)
    ckpt = os.path.join(str(tmp_path), "net_00000.pth")
    assert os.path.exists(ckpt) and open(os.path.join(str(tmp_path), "latest.ckpt")).read().strip() == "00000"      # main.py:869 echoes the zero-padded epoch
    sd = torch.load(ckpt, map_location="cpu")["network"]              # main.py:857-872 format
    ref = orc.Global_and_Local(["1"])
    ref.load_state_dict(sd, strict=True)                               # the oracle (== reference key set) accepts it
    out = t.eval(net_path=ckpt)
    assert set(out) == {"1"} and all(np.isfinite(x) for x in out["1"])



_ORACLE_TRAJ = []


@pytest.mark.parametrize("prec", ["f32", "f16x3"])
def test_three_adam_steps_see_fresh_weights(prec):
    """Every weight-derived cache of the engine (tap-major / transposed / pre-split layouts, measured maxima) must be
    refreshed after the fused Adam has written the parameters through raw pointers: three train steps (forward, fused
    Adam, forward ...) against the oracle stepping with torch.optim.Adam, plus a direct check that a 3x3 conv and the
    attention block evaluate with the UPDATED weights."""
    from glfusion_amd import ops
    from glfusion_amd.models import Global_and_Local
    from glfusion_amd.optim import Adam
    ops.set_precision(prec)
    try:
        views, n = ["1"], 4
        ref = orc.Global_and_Local(views)
        orc.closed_form_fill(ref, salt=11)
        orc.set_dropout(ref, 0.0)
        model = Global_and_Local(views)
        model.load_state_dict(ref.state_dict(), strict=True)
        orc.set_dropout(model, 0.0)
        model = model.to(DEV).train()
        ref.train()
        imgs = orc.closed_form_images(views, n, 112, 112)
        tgts = orc.closed_form_targets(views, n)
        dimgs = {v: t.to(DEV) for v, t in imgs.items()}
        dtgts = {v: t.to(DEV) for v, t in tgts.items()}
        lr = 1e-3
        opt_ref = torch.optim.Adam([p for nm, p in ref.named_parameters() if not nm.startswith("network.")], lr=lr, weight_decay=1e-5)
        opt = Adam([p for nm, p in model.named_parameters() if not nm.startswith("network.")], lr=lr, weight_decay=1e-5)
        want, got = _ORACLE_TRAJ, []
        for step in range(3):
            if len(want) <= step:                         # the oracle's three steps run once for both precisions
                pred = ref(imgs)[0]
                l = sum(torch.nn.functional.binary_cross_entropy_with_logits(pred[v], tgts[v], reduction="sum") for v in views)
                opt_ref.zero_grad()
                l.backward()
                opt_ref.step()
                want.append(float(l.detach()))
            pred = model(dimgs)[0]
            l = sum(ops.bce_with_logits_sum(pred[v], dtgts[v]) for v in views)
            opt.zero_grad(set_to_none=True)
            l.backward()
            opt.step()
            got.append(float(l))
            if step == 0:
                # direct: a 3x3 conv (tap-major cache) and a dilated one evaluate with the weights Adam just wrote
                for conv, cin, hw in ((model.layer1["1"][0].conv2, 64, 20), (model.layer4["1"][1].conv2, 512, 12)):
                    x = orc.closed_form_tensor((2, cin, hw, hw), 900 + cin, -1.0, 1.0)
                    with torch.no_grad():
                        y = conv(x.to(DEV)).cpu()
                        y_ref = torch.nn.functional.conv2d(x, conv.weight.detach().cpu(), None, conv.stride, conv.padding, conv.dilation)
                    assert float((y - y_ref).abs().max()) <= 1e-4 * float(y_ref.abs().max()), "stale 3x3 weights after Adam.step()"
                    # dgrad uses the cached transposed layout
                    xg = x.to(DEV).requires_grad_(True)
                    gy = orc.closed_form_tensor(tuple(y.shape), 901, -1.0, 1.0)
                    conv(xg).backward(gy.to(DEV))
                    xr = x.clone().requires_grad_(True)
                    torch.nn.functional.conv2d(xr, conv.weight.detach().cpu(), None, conv.stride, conv.padding, conv.dilation).backward(gy)
                    assert float((xg.grad.cpu() - xr.grad).abs().max()) <= 1e-4 * float(xr.grad.abs().max()), "stale dgrad weights after Adam.step()"
                    conv.weight.grad = None
        print("loss trajectory", got, "oracle", want)
        assert abs(got[0] - want[0]) <= 2e-5 * abs(want[0])
        # later steps: Adam's first updates are sign-like (lr * g / |g|), so every entry whose gradient is rounding noise
        # moves by the full lr either way and two fp32 evaluations separate quickly at this (deliberately large) learning
        # rate -- measured: exact-fp32 engine vs oracle 3e-3 after one update, 9e-3 after two, the split-fp16 kernels 2e-3 /
        # 2.4e-2.  The direct checks above are what pins the caches; the trajectory only has to stay in that band.
        for (a, b), tol in zip(zip(got[1:], want[1:]), (1.5e-2, 6e-2)):
            assert abs(a - b) <= tol * abs(b), (got, want)
    finally:
        ops.set_precision("f32")


def _small_model(views=("1",), salt=3, dropout=0.0):
    from glfusion_amd.models import Global_and_Local
    ref = orc.Global_and_Local(list(views))
    orc.kinkfree_fill(ref, salt=salt)         # no ReLU input near zero: two runs cannot differ by a flipped mask
    model = Global_and_Local(list(views))
    model.load_state_dict(ref.state_dict(), strict=True)
    if dropout is not None:
        orc.set_dropout(model, dropout)
    return model.to(DEV).train()


def test_weights_refresh_is_bit_identical_to_per_tensor_rebuild():
    """ops.refresh_weights() (glf_weights_refresh: every weight-derived image in four launches) against the per-tensor cache
    misses it replaces: same buffers, bit-identical contents, caches current afterwards."""
    from glfusion_amd import ops
    ops.set_precision("f16x3")
    try:
        views, n = ["1"], 2
        model = _small_model(views)
        imgs = {v: t.to(DEV) for v, t in orc.closed_form_images(views, n, 112, 112).items()}
        tgts = {v: t.to(DEV) for v, t in orc.closed_form_targets(views, n).items()}

        def step():
            for p in model.parameters():
                p.grad = None
            pred = model(imgs)[0]
            sum(ops.bce_with_logits_sum(pred[v], tgts[v]) for v in views).backward()

        step()                                             # registers every image this model's step uses
        reg = ops._registry(torch.device(DEV, torch.cuda.current_device()))
        mine = {id(p) for p in model.parameters()}
        # this model's images: derived from one of its parameters directly or through a stacked operand assembled from them (models of
        # earlier tests that are still alive keep theirs registered -- e.g. the bf16 images of a 16-bit-storage trainer)
        keys = [k for k, im in reg.images.items() if ops._image_belongs(im, mine)]
        kinds = {reg.images[k].kind for k in keys}
        assert kinds == {0, 1, 2, 3, 4, 5}, kinds          # copy, amax, both tap-major forms, transpose, packed
        assert len(keys) > 300
        with torch.no_grad():
            for i, p in enumerate(model.parameters()):
                p.mul_(1.0 + 0.01 * ((i % 7) - 3))          # new values, version counters bumped
        ops.refresh_weights()
        torch.cuda.synchronize()
        snap, ptrs = {}, {}
        for k in keys:
            im = reg.images[k]
            buf = im.amax if im.kind == 1 else im.dst
            snap[k], ptrs[k] = buf.clone(), buf.data_ptr()
            o = im.owner()
            assert im.version == ops._wversion(o), "cache not marked current by the refresh"
        # the same values again through the per-tensor path: mark every parameter changed, let the caches miss one by one
        torch.autograd.graph.increment_version(list(model.parameters()))
        step()
        torch.cuda.synchronize()
        for k in keys:
            im = reg.images[k]
            buf = im.amax if im.kind == 1 else im.dst
            assert buf.data_ptr() == ptrs[k], "an image moved: a captured refresh would write a dead buffer"
            assert torch.equal(buf.view(torch.int32), snap[k].view(torch.int32)), f"image {k[1]} (kind {im.kind}) differs"
    finally:
        ops.set_precision("f32")


@pytest.mark.parametrize("prec", ["f32", "f16x3", "bf16"])
def test_step_graph_replay_matches_eager_step(prec):
    """engine.StepGraph: the step recorded as ONE hipGraph gives the eager step's loss and gradients; a replay after the fused
    Adam wrote the parameters sees the new weights (the weight-image refresh is part of the recorded work); Dropout draws a new
    mask on every replay."""
    from glfusion_amd import ops
    from glfusion_amd.engine import StepGraph
    from glfusion_amd.optim import Adam
    ops.set_precision(prec)
    try:
        views, n = ["1"], 8                  # (8 frames: the ASPP pooled branch's BatchNorm over N frame averages is well-conditioned)
        imgs = {v: t.to(DEV) for v, t in orc.varied_images(views, n).items()}
        tgts = {v: t.to(DEV) for v, t in orc.closed_form_targets(views, n).items()}
        lr = 1e-4
        names = []

        def make(dropout):
            model = _small_model(views, salt=4, dropout=dropout)
            params = [p for nm, p in model.named_parameters() if not nm.startswith("network.")]
            names[:] = [nm for nm, p in model.named_parameters() if not nm.startswith("network.")]

            def core():
                pred = model(imgs)[0]
                loss = sum(ops.bce_with_logits_sum(pred[v], tgts[v]) for v in views)
                loss.backward()
                return loss.detach()
            return model, params, core

        # eager trajectory: step, Adam, step
        model, params, core = make(0.0)
        opt = Adam(params, lr=lr)
        want = []
        for _ in range(2):
            for p in params:
                p.grad = None
            want.append((float(core()), {i: p.grad.clone() for i, p in enumerate(params) if p.grad is not None}))
            opt.step()
        del model, opt
        # the same through the graph
        model, params, core = make(0.0)
        opt = Adam(params, lr=lr)
        sg = StepGraph(core, params, warmup=2)
        for k in range(2):
            loss = float(sg.replay())
            # (after an Adam update the two runs' rounding noise has been amplified by its sign-like first step)
            # Adam's first update moves every weight by +-lr whatever its gradient's size, so entries whose gradient is rounding
            # noise move in run-dependent directions and the second steps of two RUNS separate; k = 1 only has to show that the
            # replay saw the updated weights (the loss moved by far more than the tolerance) -- step 0 is the tight comparison
            # (bf16: two runs of the same step differ where an f64-atomic sum's last bit flips a bf16 rounding downstream -- the
            # replay is compared at the mode's own run-to-run noise, not at fp32's)
            assert abs(loss - want[k][0]) <= ((2e-4 if prec == "bf16" else 1e-6) if k == 0 else 1e-3) * abs(want[k][0]), (k, loss, want[k][0])
            if k == 1:
                assert abs(want[1][0] - want[0][0]) > 5e-3 * abs(want[0][0]), "the test's update is too small to tell stale weights"
                break
            # floor: a conv bias in front of a train-mode BatchNorm has a structurally zero gradient -- what two runs hold
            # there is rounding noise of the float-atomic ASPP rectangles, different from run to run in eager mode too
            floor = 2e-5 * max(float(g.norm()) for g in want[k][1].values())
            for i, p in enumerate(params):
                if i in want[k][1]:
                    ref = want[k][1][i]
                    err = float((p.grad - ref).norm())
                    # (the ASPP pooled branch normalises N = 4 frame averages per channel: ill-conditioned -- the float-atomic
                    # noise of two RUNS of the same eager step already moves its gradients by ~1e-3)
                    rel = 1e-2 if ".convs.4." in names[i] else 1e-3
                    if prec == "bf16":
                        rel = 0.5 if ".convs.4." in names[i] else 5e-2
                    assert err <= rel * float(ref.norm()) + floor, (k, names[i], err, float(ref.norm()), floor)
                else:
                    assert p.grad is None
            opt.zero_grad(set_to_none=True)
            sg.replay()                                    # gradients come back after set_to_none
            assert all(p.grad is not None for i, p in enumerate(params) if i in want[k][1])
            opt.step()
        sg.release()
        del sg, model, opt
        # Dropout(0.5) active: every replay draws its own mask
        model, params, core = make(None)
        sg = StepGraph(core, params, warmup=1)
        losses = [float(sg.replay()) for _ in range(3)]
        assert len({round(l, 3) for l in losses}) == 3, losses
        sg.release()
    finally:
        ops.set_precision("f32")


def test_trainer_graph_mode_follows_the_eager_trainer(tmp_path):
    """config['train']['graph'] = True: Trainer.train_step replays the segmentation step from one hipGraph (static input buffers,
    fused Adam in place, weight images refreshed inside the recorded work).  Three steps on three different batches against the
    eager Trainer started from the same weights: same losses (Dropout off), the parameters move together."""
    from glfusion_amd.engine import Trainer
    from glfusion_amd import ops
    ops.set_precision("f16x3")
    try:
        def make(graph):
            cfg = {"train": {"view_num": ["1"], "test_view": ["1"], "num_epochs": 1, "batch_size": 4, "iters_per_epoch": 3, "clip_length": 8,
                             "save_dir": str(tmp_path), "validate_every_epoch": False, "graph": graph},
                   "net": {"opt": {"opt_name": "Adam", "lr": 1e-5, "weight_decay": 1e-5}}}
            torch.manual_seed(0)
            t = Trainer(cfg)
            orc.kinkfree_fill(t.model, salt=9)
            orc.set_dropout(t.model, 0.0)
            t.model.train()
            return t
        eager, graph = make(False), make(True)
        batches = [eager.loader.batch() for _ in range(3)]
        le = [float(eager.train_step(i, m)[0]) for i, m in batches]
        lg = [float(graph.train_step(i, m)[0]) for i, m in batches]
        assert graph._graph is not None
        for a, b in zip(le, lg):
            assert abs(a - b) <= 1e-4 * abs(a), (le, lg)
        assert len({round(x, 1) for x in lg}) == 3                      # three different batches went through the static buffers
        pe, pg = dict(eager.model.named_parameters()), dict(graph.model.named_parameters())
        k = "layer4.1.2.conv3.weight"
        assert float((pe[k] - pg[k]).norm()) <= 5e-3 * float((pe[k]).norm())
        moved = float((pg[k] - make(False).model.state_dict()[k]).norm())
        assert moved > 0.0
    finally:
        ops.set_precision("f32")
