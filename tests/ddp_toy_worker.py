"""One rank of the CPU rehearsal of bench.py's self-launch plumbing (tests/test_ddp_cpu.py::test_self_launch_world8): reads
RANK / WORLD_SIZE / MASTER_* from the environment exactly as bench.py does, joins a gloo group, runs a toy model through
glfusion_amd.ddp.GradAllReducer (immediate and deferred mode) and -- on rank 0 -- prints ONE JSON line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
assert int(os.environ["LOCAL_RANK"]) == rank and os.environ["MASTER_ADDR"] == "127.0.0.1"
dist.init_process_group("gloo", rank=rank, world_size=world)
from glfusion_amd.ddp import GradAllReducer, shard_frames

torch.manual_seed(100 + rank)
model = torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.Tanh(), torch.nn.Linear(16, 3))
red = GradAllReducer(model, bucket_mb=0.0001)
red.broadcast_parameters(0)
one = torch.ones(1)
dist.all_reduce(one)
g = torch.Generator().manual_seed(7)
x, t = torch.randn(8 * world, 6, generator=g), torch.randn(8 * world, 3, generator=g)
lo, hi = shard_frames(8 * world, rank, world)
out = {}
for mode in ("immediate", "deferred"):
    red.deferred = mode == "deferred"
    for p in model.parameters():
        p.grad = None
    ((model(x[lo:hi]) - t[lo:hi]) ** 2).sum().backward()
    red.finalize()
    out[mode] = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
# every rank must hold the gradient of the SUM loss over the global batch
ref = torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.Tanh(), torch.nn.Linear(16, 3))
ref.load_state_dict(model.state_dict())
((ref(x) - t) ** 2).sum().backward()
truth = torch.cat([p.grad.reshape(-1) for p in ref.parameters()])
err = max(float((out[m] - truth).abs().max()) for m in out)
worst = torch.tensor([err])
dist.all_reduce(worst, op=dist.ReduceOp.MAX)
if rank == 0:
    print(json.dumps({"n_gpus": world, "ranks_in_collective": int(one.item()), "max_abs_err": float(worst), "buckets": len(red.buckets),
                      "argv": sys.argv[1:]}), flush=True)
dist.barrier()
dist.destroy_process_group()
