import sys, time; sys.path.insert(0,'/root/repo')
import torch
from glfusion_amd import ops
from glfusion_amd.models import Global_and_Local
ops.set_precision(sys.argv[2] if len(sys.argv)>2 else 'f16x3')
dev=torch.device('cuda',0)
views=['1','2','3','4','5']; T=int(sys.argv[1]) if len(sys.argv)>1 else 32; H=W=224
torch.manual_seed(0)
model=Global_and_Local(views)
with torch.no_grad():
    for a in (model.global_attn, model.local_attn): a.W_z[1].weight.normal_(1.0,0.1)
model=model.to(dev).train()
g=torch.Generator(device=dev).manual_seed(1)
imgs={v: torch.rand(T,1,H,W,device=dev,generator=g) for v in views}
tg={v: (torch.rand(T,5,H,W,device=dev,generator=g)<0.3).float() for v in views}
def step():
    for p in model.parameters(): p.grad=None
    pred=model(imgs)[0]
    loss=None
    for v in views:
        l=ops.bce_with_logits_sum(pred[v],tg[v]); loss=l if loss is None else loss+l
    loss.backward(); return loss
step(); torch.cuda.synchronize()
t0=time.perf_counter(); l=step(); torch.cuda.synchronize(); dt=time.perf_counter()-t0
print(f"config-5 shape: 5 views x {T} x 224^2, L={5*56*56}: {dt*1e3:.0f} ms/step = {1/dt:.3f} clips/s, loss {float(l):.1f}, peak mem {torch.cuda.max_memory_allocated()/2**30:.1f} GB; dense 504.6 TFLOP/clip -> {504.6*T/32/dt:.0f} TFLOP/s dense-equivalent")
