"""CPU, world_size 2, gloo: the gradient all-reduce wrapper (glfusion_amd.ddp) produces on every rank
the gradient of the SUM loss over the global batch, identical parameters stay identical, unused
parameters are tolerated, and the frame sharding is disjoint and complete."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import glfusion_ref as orc


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class _Tiny(torch.nn.Module):
    """Same ingredients as the path at toy size (conv/BN/TPAVI + a dead `network` template)."""

    def __init__(self):
        super().__init__()
        self.network = torch.nn.Linear(4, 4)                 # never used: no gradient
        self.stem = torch.nn.Conv2d(1, 8, 3, padding=1)
        self.bn = torch.nn.BatchNorm2d(8)
        self.attn = orc.TPAVIModule(8, mode="dot")
        self.head = torch.nn.Conv2d(8, 5, 1)
        self.spare = torch.nn.Conv2d(8, 1, 1)                # registered, NOT ignored, never used (e.g. Global_only's centerness heads)

    def forward(self, x):
        f = torch.relu(self.bn(self.stem(x)))
        z, _ = self.attn(torch.stack([f, f * 0.5], dim=2))
        return self.head(z[:, :, 0] + z[:, :, 1])


def _worker(rank, world, port, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from glfusion_amd.ddp import GradAllReducer, shard_frames, all_reduce_counts
    torch.manual_seed(1234 + rank)                            # different init per rank on purpose
    model = _Tiny()
    red = GradAllReducer(model, bucket_mb=0.0002)              # tiny buckets => several collectives in flight
    assert len(red.buckets) > 2
    assert all(not n.startswith("network.") and "align_channel" not in n for n in red.names)
    red.broadcast_parameters(0)
    n_total = 8
    xs = orc.closed_form_tensor((n_total, 1, 12, 12), 5)
    ts = (orc.closed_form_tensor((n_total, 5, 12, 12), 6) < 0.3).float()
    lo, hi = shard_frames(n_total, rank, world)
    bce = torch.nn.BCEWithLogitsLoss(reduction="sum")
    model.eval()                                              # eval BN => per-sample independence => exact check
    loss = bce(model(xs[lo:hi]), ts[lo:hi])
    loss.backward()
    red.finalize()
    # a bucketed parameter that received no gradient keeps .grad None (the optimizer must skip it as on one GPU)
    assert "spare.weight" in red.names and model.spare.weight.grad is None and model.spare.bias.grad is None
    grads = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    counts = all_reduce_counts(torch.tensor([rank + 1, 2, 3, 4]))
    torch.save({"grads": grads, "state": model.state_dict(), "range": (lo, hi), "counts": counts}, os.path.join(tmp, f"r{rank}.pt"))
    if rank == 0:                                             # single-process truth on the whole batch
        ref = _Tiny()
        ref.load_state_dict(model.state_dict())
        ref.eval()
        bce(ref(xs), ts).backward()
        torch.save({n: p.grad for n, p in ref.named_parameters() if p.grad is not None}, os.path.join(tmp, "truth.pt"))
    # second step: hooks re-arm, grads keep matching across ranks
    for p in model.parameters():
        p.grad = None
    bce(model(xs[lo:hi]), ts[lo:hi]).backward()
    red.finalize()
    torch.save({n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}, os.path.join(tmp, f"s{rank}.pt"))
    # finalize() has waited for every collective it launched (on a GPU: the stream the optimizer runs on waits for them) and re-armed
    assert all(b.work is None and b.pending == len(b.params) and not any(b.fired) for b in red.buckets)
    want = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    # an eager step that DOES use `spare` leaves reduced values in its bucket slices; the deferred steps that
    # follow do not use it -- its slices must contribute zeros (not the stale sums, re-reduced in place on every replay), and going
    # back to eager mode afterwards must work from clean per-step state
    for p in model.parameters():
        p.grad = None
    model.spare(torch.ones(1, 8, 2, 2)).sum().backward()
    red.finalize()
    assert model.spare.weight.grad is not None and float(model.spare.weight.grad.abs().sum()) > 0
    wb, wi = red._where[model.spare.weight]
    red.deferred = True
    for _ in range(3):
        for p in model.parameters():
            p.grad = None
        bce(model(xs[lo:hi]), ts[lo:hi]).backward()
        red.finalize()
        off = wb.offsets[wi]
        assert float(wb.flat[off:off + model.spare.weight.numel()].abs().sum()) == 0.0
        assert model.spare.weight.grad is None
        for n, p in model.named_parameters():
            if n in want:
                assert torch.allclose(p.grad, want[n], rtol=1e-5, atol=1e-6), ("deferred", n)
    red.deferred = False
    for p in model.parameters():
        p.grad = None
    bce(model(xs[lo:hi]), ts[lo:hi]).backward()
    red.finalize()
    for n, p in model.named_parameters():
        if n in want:
            assert torch.allclose(p.grad, want[n], rtol=1e-5, atol=1e-6), ("eager after deferred", n)
    assert model.spare.weight.grad is None
    dist.destroy_process_group()


def test_grad_allreduce_world2(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(tmp_path / f"r{i}.pt") for i in range(world)]
    truth = torch.load(tmp_path / "truth.pt")
    assert r[0]["range"] == (0, 4) and r[1]["range"] == (4, 8)
    assert r[0]["counts"].tolist() == [3, 4, 6, 8]
    for k in r[0]["state"]:
        assert torch.equal(r[0]["state"][k], r[1]["state"][k]), k           # broadcast made replicas identical
    assert set(r[0]["grads"]) == set(truth) and not any(n.startswith("network.") for n in truth)
    for n, g in truth.items():
        assert torch.equal(r[0]["grads"][n], r[1]["grads"][n]), n            # all ranks hold the same reduced gradient
        assert torch.allclose(r[0]["grads"][n], g, rtol=1e-4, atol=1e-5), n  # == gradient of the global SUM loss
    s = [torch.load(tmp_path / f"s{i}.pt") for i in range(world)]
    for n in truth:
        assert torch.equal(s[0][n], s[1][n]) and torch.allclose(s[0][n], r[0]["grads"][n], rtol=1e-5, atol=1e-6), n


def test_shard_frames_errors():
    from glfusion_amd.ddp import shard_frames
    with pytest.raises(ValueError):
        shard_frames(10, 0, 4)
    assert [shard_frames(64, r, 8) for r in (0, 7)] == [(0, 8), (56, 64)]


def test_self_launch_world8(tmp_path):
    """bench.py's `--gpus N` self-launch plumbing at N = 8 (what the driver's 8-GPU run relies on when it starts the script
    plainly): eight rank processes with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, all of them in the collective, rank 0's
    JSON line relayed, a failing rank turned into a non-zero exit.  CPU + gloo with a toy model (tests/ddp_toy_worker.py)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    worker = os.path.join(root, "tests", "ddp_toy_worker.py")
    code = f"import sys; sys.path.insert(0, {root!r}); import bench; raise SystemExit(bench.self_launch(8, {worker!r}, ['--tag', 'x']))"
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 8 and line["ranks_in_collective"] == 8 and line["argv"] == ["--tag", "x"]
    assert line["max_abs_err"] < 1e-4 and line["buckets"] > 2
    # a rank that dies makes the launcher fail
    bad = os.path.join(str(tmp_path), "bad.py")
    open(bad, "w").write("import os, sys\nsys.exit(3 if os.environ['RANK'] == '5' else 0)\n")
    code = f"import sys; sys.path.insert(0, {root!r}); import bench; raise SystemExit(bench.self_launch(8, {bad!r}, []))"
    assert subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120).returncode != 0
