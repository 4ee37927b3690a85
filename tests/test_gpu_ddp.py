"""GPU: two ranks of the HIP engine (sharing the one GPU of the test box over gloo -- RCCL needs one device per
rank, the driver runs that at round end) exchange gradients through glfusion_amd.ddp.GradAllReducer: every rank
ends with the same gradients, and they equal the single-process gradient of the SUM loss over the global batch
(eval-mode BatchNorm => frames are independent => exact up to summation order)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

VIEWS = ["1", "3"]
N_TOTAL = 4


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _make(dev):
    from oracle import glfusion_ref as orc
    from glfusion_amd.models import Global_and_Local
    model = Global_and_Local(VIEWS)
    orc.closed_form_fill(model, salt=1)
    model = model.to(dev).eval()
    imgs = {v: t.to(dev) for v, t in orc.closed_form_images(VIEWS, N_TOTAL).items()}
    tgts = {v: t.to(dev) for v, t in orc.closed_form_targets(VIEWS, N_TOTAL).items()}
    return model, imgs, tgts


def _step(model, imgs, tgts, lo, hi):
    from glfusion_amd import ops
    pred = model({v: t[lo:hi] for v, t in imgs.items()})[0]
    loss = sum(ops.bce_with_logits_sum(pred[v], tgts[v][lo:hi]) for v in VIEWS)
    loss.backward()
    return float(loss.detach())


def _worker(rank, world, port, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from glfusion_amd.ddp import GradAllReducer, shard_frames
    dev = torch.device("cuda", 0)
    model, imgs, tgts = _make(dev)
    red = GradAllReducer(model, bucket_mb=16.0)
    red.broadcast_parameters(0)
    lo, hi = shard_frames(N_TOTAL, rank, world)
    loss = _step(model, imgs, tgts, lo, hi)
    red.finalize()
    torch.cuda.synchronize()
    sel = {n: p.grad.detach().float().cpu() for n, p in model.named_parameters()
           if p.grad is not None and (n.endswith("conv1.weight") or "attn" in n or n.endswith(".4.weight") or n.startswith("init_block"))}
    torch.save({"loss": loss, "grads": sel, "n_grads": sum(p.grad is not None for p in model.parameters())}, os.path.join(tmp, f"r{rank}.pt"))
    dist.destroy_process_group()


def test_two_ranks_match_single_process(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(tmp_path / f"r{i}.pt") for i in range(world)]
    model, imgs, tgts = _make(torch.device("cuda", 0))
    loss = _step(model, imgs, tgts, 0, N_TOTAL)
    assert abs((r[0]["loss"] + r[1]["loss"]) - loss) <= 1e-5 * abs(loss)
    assert r[0]["n_grads"] == r[1]["n_grads"] == sum(p.grad is not None for p in model.parameters())
    truth = dict(model.named_parameters())
    assert len(r[0]["grads"]) > 20
    for n, g0 in r[0]["grads"].items():
        assert torch.equal(g0, r[1]["grads"][n]), n                          # identical on every rank
        t = truth[n].grad.detach().float().cpu().double()
        err = float((g0.double() - t).norm()) / max(float(t.norm()), 1e-12)
        assert err <= 5e-4, (n, err)                                          # == gradient of the global SUM loss
