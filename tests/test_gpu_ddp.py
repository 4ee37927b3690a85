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
    # the same step recorded as ONE hipGraph (engine.StepGraph): the reducer goes into deferred mode -- the hooks' bucket copies are
    # part of the recorded work, the collectives are launched by finalize() after every replay
    from glfusion_amd.engine import StepGraph
    from glfusion_amd import ops
    params = [p for p in model.parameters()]

    def core():
        pred = model({v: t[lo:hi] for v, t in imgs.items()})[0]
        l = sum(ops.bce_with_logits_sum(pred[v], tgts[v][lo:hi]) for v in VIEWS)
        l.backward()
        return l.detach()
    sg = StepGraph(core, params, warmup=1, reducer=red)
    gl = float(sg.replay())
    torch.cuda.synchronize()
    graph_err = 0.0
    for n, p in model.named_parameters():
        if n in sel:
            graph_err = max(graph_err, float((p.grad.detach().float().cpu() - sel[n]).norm()) / max(float(sel[n].norm()), 1e-12))
    ar_ms = red.last_allreduce_ms()
    sg.release()
    torch.save({"loss": loss, "grads": sel, "n_grads": sum(p.grad is not None for p in model.parameters()), "graph_loss": gl, "graph_err": graph_err,
                "allreduce_ms": ar_ms}, os.path.join(tmp, f"r{rank}.pt"))
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
    for i in range(world):                                                    # the graph-replayed step reduces to the same gradients
        assert abs(r[i]["graph_loss"] - r[i]["loss"]) <= 1e-5 * abs(r[i]["loss"]), (r[i]["graph_loss"], r[i]["loss"])
        assert r[i]["graph_err"] <= 2e-4, r[i]["graph_err"]          # two runs of one step: float-atomic ASPP rectangles + ReLU kinks of this fill
        assert r[i]["allreduce_ms"] is not None and r[i]["allreduce_ms"] > 0


# ------------------------------------------------------------------------------------------------------------
# config 4 (BASELINE.json configs[3]): batch 32 over 8 GPUs = 4 clips of 16 frames x 3 views PER RANK.  Two such
# ranks fit the one GPU of the test box (2 x 58 GB): the per-rank workload is exactly C4's.
# ------------------------------------------------------------------------------------------------------------
C4_VIEWS = ["1", "3", "4"]
C4_FRAMES = 64                      # per view per rank: B_local = 4 clips x T = 16


def _c4_worker(rank, world, port, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import glfusion_ref as orc
    from glfusion_amd import ops
    from glfusion_amd.ddp import GradAllReducer
    from glfusion_amd.models import Global_and_Local
    ops.set_precision("f16x3")                                   # the bench's default contraction kernels
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    model = Global_and_Local(C4_VIEWS)
    with torch.no_grad():
        for attn in (model.global_attn, model.local_attn):
            attn.W_z[1].weight.normal_(1.0, 0.1)
    orc.set_dropout(model, 0.0)                                  # two identical passes below
    model = model.to(dev).train()
    g = torch.Generator(device=dev).manual_seed(1234 + rank)     # bench.py's per-rank synthetic shard
    imgs = {v: torch.rand(C4_FRAMES, 1, 112, 112, device=dev, generator=g) for v in C4_VIEWS}
    tgts = {v: (torch.rand(C4_FRAMES, 5, 112, 112, device=dev, generator=g) < 0.3).float() for v in C4_VIEWS}
    keep = lambda n: n.endswith("layer4.1.0.conv2.weight") or n.startswith("global_attn.theta") or n.endswith("classifier.3.4.weight") or n.startswith("init_block.4.0")

    def step():
        for p in model.parameters():
            p.grad = None
        pred = model(imgs)[0]
        loss = sum(ops.bce_with_logits_sum(pred[v], tgts[v]) for v in C4_VIEWS)
        loss.backward()
        return float(loss.detach())

    loss_local = step()                                          # 1: a pass without exchange (loss reproducibility)
    # this rank's OWN gradients of the reducing pass, captured by hooks that run before the reducer's (registration order)
    local = {}
    for n, p in model.named_parameters():
        if keep(n):
            p.register_post_accumulate_grad_hook(lambda q, n=n: local.__setitem__(n, q.grad.detach().cpu().clone()))
    red = GradAllReducer(model)                                  # default 48 MB buckets, as bench.py / engine.Trainer
    red.broadcast_parameters(0)
    loss = step()                                                # 2: same batch, hooks armed
    red.finalize()
    torch.cuda.synchronize()
    reduced = {n: p.grad.detach().cpu().clone() for n, p in model.named_parameters() if p.grad is not None and keep(n)}
    torch.save({"loss": loss, "loss_local": loss_local, "local": local, "reduced": reduced,
                "in_place": red.in_place_elems, "copied": red.copied_elems, "overlap": list(red.overlap_log),
                "n_grads": sum(p.grad is not None for p in model.parameters()), "peak_gb": torch.cuda.max_memory_allocated() / 2 ** 30},
               os.path.join(tmp, f"c4_{rank}.pt"))
    dist.destroy_process_group()


def test_config4_two_ranks_at_c2_frame_counts(tmp_path):
    """Each of two ranks runs the full C2-sized shard (3 views x 64 frames, train mode) and exchanges gradients: the
    reduced gradient is the SUM of the ranks' own gradients, identical on both ranks; losses are finite."""
    world = 2
    mp.spawn(_c4_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(tmp_path / f"c4_{i}.pt") for i in range(world)]
    for i in range(world):
        assert r[i]["loss"] == r[i]["loss"] and abs(r[i]["loss"]) < 1e12
        assert abs(r[i]["loss"] - r[i]["loss_local"]) <= 1e-5 * abs(r[i]["loss"])     # same batch, same batch statistics
    assert r[0]["n_grads"] == r[1]["n_grads"] > 600
    assert len(r[0]["reduced"]) >= 4
    for n, g0 in r[0]["reduced"].items():
        assert torch.equal(g0, r[1]["reduced"][n]), n
        want = r[0]["local"][n].double() + r[1]["local"][n].double()
        assert float((g0.double() - want).norm()) <= 1e-6 * float(want.norm()), n     # a two-term fp32 sum: order-free
    for x in r:
        # gradients of parameters used ONCE per forward are produced INSIDE their bucket slice (conv / linear weights, BatchNorm
        # gamma / beta of the encoders and the centerness heads: 64 % of the elements) -- no copy; what is still copied: the
        # classifier heads (applied three times per forward: autograd sums the contributions out of place) and the fusion blocks'
        # stacked projection gradients
        assert x["in_place"] > 0.6 * (x["in_place"] + x["copied"]), (x["in_place"], x["copied"])
        # and the collectives of all but the last bucket were enqueued while backward was still producing gradients
        assert len(x["overlap"]) >= 3 and sum(fired < total for _, fired, total in x["overlap"]) >= len(x["overlap"]) - 1, x["overlap"]
    print("config-4 rank peak memory (GB):", [round(x["peak_gb"], 1) for x in r], "in-place gradient elements:",
          [round(x["in_place"] / (x["in_place"] + x["copied"]), 3) for x in r])


def test_bench_self_launch_two_ranks():
    """`python bench.py --gpus 2` from a plain shell (no torch.distributed.run): the script starts its own rank
    processes, both join the collective, rank 0 prints the JSON line.  gloo here (one GPU on the test box; RCCL wants
    one device per rank), one clip per rank to keep it short."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # the multi-rank default: the eager step, collectives overlapped with backward (what the driver's multi-GPU run executes).  The
    # replayed step under a multi-rank reducer (GLF_BENCH_GRAPH=1) is covered by test_two_ranks_match_single_process.
    for graph in ("0",):
        env = dict(os.environ, GLF_DIST_BACKEND="gloo", GLF_BENCH_GRAPH=graph)
        env.pop("WORLD_SIZE", None)
        env.pop("RANK", None)
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--clips", "1",
                              "--no-exact-f32", "--no-config3", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stderr[-2000:]
        line = json.loads(out.stdout.strip().splitlines()[-1])
        assert line["n_gpus"] == 2 and line["config"]["ranks_in_collective"] == 2
        assert line["config"]["global_batch_clips"] == 2 and line["value"] > 0 and line["scaling"] == "weak"
        assert ("hipGraph" in line["launch"]) == (graph == "1")
        if graph == "1":
            assert line["config"]["allreduce_ms_per_step"] is not None


def _trainer_worker(rank, world, port, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from glfusion_amd.engine import Trainer
    cfg = {"train": {"batch_size": 2, "num_epochs": 2, "clip_length": 8, "view_num": ["1"], "test_view": ["1"], "dense_cyc": False,
                     "save_dir": os.path.join(tmp, f"ckpt{rank}"), "iters_per_epoch": 1, "global_rank": rank, "validate_every_epoch": True},
           "net": {"opt": {"opt_name": "Adam", "lr": 1e-5, "params": (0.9, 0.999), "weight_decay": 1e-5}}}
    t = Trainer(cfg)
    assert t.print_val == (rank == 0)
    t.train(is_backbone=False, is_cycle=False)                # two epochs: the second one's all-reduces follow rank 0's validation
    torch.cuda.synchronize()
    w = t.model.classifier["1"][4].weight.detach().float().cpu()
    torch.save({"w": w, "val": getattr(t, "validation_report", None) is not None}, os.path.join(tmp, f"t{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_trainer_with_per_epoch_validation(tmp_path):
    """Trainer.train() with validate_every_epoch=True on two ranks.  The per-epoch Dice is a collective every
    rank joins; the validation pass runs on the printing rank alone WITHOUT collectives (main.py:259-274) -- before the fix rank 0
    entered all-reduces no other rank joined and the next epoch's gradient buckets were paired with them.  Two epochs complete,
    only rank 0 validates, and the replicas end with identical parameters."""
    world = 2
    mp.spawn(_trainer_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(tmp_path / f"t{i}.pt") for i in range(world)]
    assert r[0]["val"] and not r[1]["val"]
    assert torch.equal(r[0]["w"], r[1]["w"])
