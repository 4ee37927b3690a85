"""Loss trajectories of a short training run under the three contraction precisions (same seeds, same data)."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from glfusion_amd import ops
from glfusion_amd.models import Global_and_Local
from glfusion_amd.optim import Adam
dev = torch.device("cuda", 0)
views, n, steps = ["1", "3", "4"], 8, 10
res = {}
for prec in ("f32", "bf16x6", "f16x3"):
    ops.set_precision(prec)
    torch.manual_seed(0)
    model = Global_and_Local(views)
    with torch.no_grad():
        for a in (model.global_attn, model.local_attn):
            a.W_z[1].weight.normal_(1.0, 0.1)
    model = model.to(dev).train()
    opt = Adam([p for k, p in model.named_parameters() if not k.startswith("network.")], lr=3e-4, weight_decay=1e-5)
    g = torch.Generator(device=dev).manual_seed(1)
    imgs = {v: torch.rand(n, 1, 112, 112, device=dev, generator=g) for v in views}
    tg = {v: (torch.rand(n, 5, 112, 112, device=dev, generator=g) < 0.3).float() for v in views}
    torch.manual_seed(123)            # dropout masks
    losses = []
    for it in range(steps):
        pred = model(imgs)[0]
        loss = sum(ops.bce_with_logits_sum(pred[v], tg[v]) for v in views)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        losses.append(float(loss))
    res[prec] = losses
    print(prec, " ".join(f"{l:.1f}" for l in losses))
for prec in ("bf16x6", "f16x3"):
    print(prec, "max relative deviation from f32 per step:", " ".join(f"{abs(a - b) / abs(b):.1e}" for a, b in zip(res[prec], res["f32"])))
