"""Run-to-run spread of the exact-fp32 smoke step: the oracle's float64 gradients once, then the HIP step N times; per run the worst
relative-L2 deviation and a few named tensors.  Usage: smoke_spread.py [runs] [precision]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from glfusion_amd import ops
from glfusion_amd.models import Global_and_Local
from oracle import glfusion_ref as orc

if os.environ.get("SPREAD_NORECT"):          # no per-tap rectangle launches (the float-atomic form) for: all | fwd | dgrad | wgrad
    which = os.environ["SPREAD_NORECT"]
    for k in ops.RECT_THRESHOLD:
        if k != "region" and (which in ("1", "all") or k.startswith(which)):
            ops.RECT_THRESHOLD[k] = 0.0
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 5
prec = sys.argv[2] if len(sys.argv) > 2 else "f32"
dev = torch.device("cuda:0")
views, n = ["1"], 8
torch.set_num_threads(16)
ref = orc.Global_and_Local(views)
orc.kinkfree_fill(ref, salt=1)
orc.set_dropout(ref, 0.0)
imgs = orc.varied_images(views, n)
tgts = orc.closed_form_targets(views, n)
sd0 = {k: v.clone() for k, v in ref.state_dict().items()}
ref = ref.double().train()
orc.train_step(ref, {v: t.double() for v, t in imgs.items()}, {v: t.double() for v, t in tgts.items()})
want = {name: q.grad.double() for name, q in ref.named_parameters() if q.grad is not None}
watch = ["layer1.1.1.conv1.weight", "global_attn.theta.weight", "local_attn.theta.weight", "centerness.1.0.convs.4.1.weight", "layer4.1.2.conv3.weight"]
prev = None
for r in range(runs):
    with ops.precision_scope(prec):
        model = Global_and_Local(views)
        model.load_state_dict(sd0, strict=True)
        orc.set_dropout(model, 0.0)
        model = model.to(dev).train()
        pred = model({v: t.to(dev) for v, t in imgs.items()})[0]
        loss = sum(ops.bce_with_logits_sum(pred[v], tgts[v].to(dev)) for v in views)
        loss.backward()
        torch.cuda.synchronize()
    got = {k: p.grad.cpu().double() for k, p in model.named_parameters() if k in want}
    errs = {k: float((got[k] - want[k]).norm()) / max(float(want[k].norm()), 1e-30) for k in want}
    big = {k: e for k, e in errs.items() if float(want[k].norm()) > 1e-4 * max(float(v.norm()) for kk, v in want.items() if kk.split(".")[0] == k.split(".")[0])}
    worst = max(big.items(), key=lambda kv: kv[1])
    n_over = sum(1 for e in big.values() if e > 2e-3)
    line = f"run {r}: worst {worst[0]} {worst[1]:.2e}; tensors over 2e-3: {n_over}; " + ", ".join(f"{w.split('.')[0]}..{w.split('.')[-2]} {errs[w]:.2e}" for w in watch)
    if prev is not None:
        moved = max(float((got[k] - prev[k]).norm()) / max(float(want[k].norm()), 1e-30) for k in big)
        line += f"; largest run-to-run change {moved:.2e}"
    print(line, flush=True)
    prev = got
    del model
