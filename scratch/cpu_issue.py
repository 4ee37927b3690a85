import sys, time, os; sys.path.insert(0,'/root/repo')
import torch, bench
from glfusion_amd import ops
dev=torch.device('cuda',0)
ops.set_precision('f16x3')
model=bench.build_model(dev)
imgs,tgts=bench.make_batch(dev,0,64)
def step():
    for p in model.parameters(): p.grad=None
    pred=model(imgs)[0]
    loss=None
    for v in bench.VIEWS:
        l=ops.bce_with_logits_sum(pred[v],tgts[v]); loss=l if loss is None else loss+l
    loss.backward()
for _ in range(2): step()
torch.cuda.synchronize()
for _ in range(3):
    t0=time.perf_counter(); step(); t1=time.perf_counter(); torch.cuda.synchronize(); t2=time.perf_counter()
    print(f"cpu issue {1e3*(t1-t0):.1f} ms, total {1e3*(t2-t0):.1f} ms")
