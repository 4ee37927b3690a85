import sys; sys.path.insert(0, '/root/repo')
import torch, numpy as np
from oracle import glfusion_ref as orc
from glfusion_amd import ops
from glfusion_amd.models import DeepLabHead
DEV='cuda'
def err(a,b):
    a=a.detach().cpu().double(); b=b.detach().cpu().double()
    return "%.3e / %.3e" % (float((a-b).abs().max()), float(b.abs().max()))
ho = orc.DeepLabHead(64,5); orc.closed_form_fill(ho, salt=5); orc.set_dropout(ho,0.0); ho.train()
hh = DeepLabHead(64,5); hh.load_state_dict(ho.state_dict()); orc.set_dropout(hh,0.0); hh=hh.to(DEV).train()
h64 = orc.DeepLabHead(64,5); orc.closed_form_fill(h64, salt=5); orc.set_dropout(h64,0.0); h64=h64.double().train()
x = orc.closed_form_tensor((4,64,28,28),201,0.0,1.0)
w = None
res={}
for tag, head, xx in (('o',ho,x.clone()),('h',hh,x.to(DEV)),('d',h64,x.double())):
    xx.requires_grad_(True)
    aspp=head[0]
    brs=[b(xx) for b in aspp.convs]
    for b in brs: b.retain_grad()
    y=head(xx)
    if w is None: w=orc.closed_form_tensor(tuple(y.shape),202,-1.0,1.0)
    (y*w.to(y.device).to(y.dtype)).sum().backward()
    res[tag]=dict(y=y, dx=xx.grad, brs=brs, grads={n:p.grad for n,p in head.named_parameters()})
for a,bname in (('h','hip'),('o','cpu32')):
    print(bname,'y',err(res[a]['y'],res['d']['y']),'dx',err(res[a]['dx'],res['d']['dx']))
    for i in range(5): print('  branch',i,err(res[a]['brs'][i],res['d']['brs'][i]))
    for n in res['d']['grads']:
        print('  ',n,err(res[a]['grads'][n],res['d']['grads'][n]))
