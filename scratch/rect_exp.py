import sys; sys.path.insert(0,'/root/repo')
import torch
from glfusion_amd import ops
DEV='cuda'
def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/iters
for prec in ('f16x3',):
  ops.set_precision(prec)
  for dil in (12,24):
    x=torch.randn(64,28,28,2048,device=DEV,requires_grad=True)
    w=torch.randn(256,2048,3,3,device=DEV,requires_grad=True)*0.01
    w=w.detach().requires_grad_(True)
    for mode in ('rect/rect','dense/rect','rect/dense','dense/dense'):
        a,b=mode.split('/')
        ops.RECT_THRESHOLD.update(fwd=0.8,dgrad=0.8 if a=='rect' else 0.0,wgrad=0.8 if b=='rect' else 0.0)
        y=ops.conv2d(x,w,None,1,dil,dil)
        dy=torch.randn_like(y)
        t_f=timeit(lambda: ops.conv2d(x,w,None,1,dil,dil))
        def bwd_x():
            torch.autograd.grad(y,[x],dy,retain_graph=True)
        def bwd_w():
            torch.autograd.grad(y,[w],dy,retain_graph=True)
        t_x=timeit(bwd_x); t_w=timeit(bwd_w)
        print(f"{prec} dil {dil} {mode}: fwd {t_f:.3f} ms dgrad {t_x:.3f} ms wgrad {t_w:.3f} ms")
