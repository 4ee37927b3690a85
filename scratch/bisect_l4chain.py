import sys; sys.path.insert(0, '/root/repo')
import torch, numpy as np, copy
from oracle import glfusion_ref as orc
from glfusion_amd import ops
from glfusion_amd.models import resnet as hip_resnet
DEV='cuda'
def rel(a,t):
    a=a.detach().cpu().double(); t=t.detach().cpu().double()
    return float((a-t).abs().max())/max(float(t.abs().max()),1e-30)
trunk_o = orc.ResNet50Trunk((False, True, True)); orc.closed_form_fill(trunk_o, salt=9)
trunk_h = hip_resnet.ResNet((3,4,6,3),(False,True,True)); trunk_h.load_state_dict(trunk_o.state_dict()); trunk_h=trunk_h.to(DEV)
trunk_d = copy.deepcopy(trunk_o).double()
x = orc.closed_form_tensor((2,1024,12,12),300,0.0,1.0)
res={}
for tag,layer,xx in (('o',trunk_o.layer4.train(),x.clone()),('d',trunk_d.layer4.train(),x.double()),('h',trunk_h.layer4.train(),x.to(DEV))):
    xx.requires_grad_(True)
    acts=[xx]
    for blk in layer:
        y=blk(acts[-1]); y.retain_grad(); acts.append(y)
    gy=orc.closed_form_tensor(tuple(y.shape),301,-1.0,1.0).to(y.device).to(y.dtype)
    y.backward(gy)
    res[tag]=acts
for i in range(4):
    print('act',i,'val %.1e/%.1e'%(rel(res['h'][i],res['d'][i]),rel(res['o'][i],res['d'][i])),'grad %.1e/%.1e'%(rel(res['h'][i].grad,res['d'][i].grad),rel(res['o'][i].grad,res['d'][i].grad)))
# now via Stage.forward (NHWC chained inside)
xh=x.to(DEV).requires_grad_(True)
for p in trunk_h.parameters(): p.grad=None
yh=trunk_h.layer4(xh); yh.backward(gy.to(DEV).float())
print('stage chained dx %.1e'%rel(xh.grad,res['d'][0].grad), 'y %.1e'%rel(yh,res['d'][3]))
for i in (2,1,0):
    e=(res['h'][i].grad.detach().cpu().double()-res['d'][i].grad).abs(); m=float(res['d'][i].grad.abs().max())
    print('act',i,'n elems', e.numel(), 'n err>1e-4*max:', int((e>1e-4*m).sum()), 'n err>1e-5*max:', int((e>1e-5*m).sum()), 'L2 rel %.2e'%(float(e.norm())/float(res['d'][i].grad.norm())))
# mask flips at block outputs
for i in (1,2,3):
    fl=((res['h'][i].detach().cpu()>0)!=(res['d'][i].detach()>0)); print('act',i,'relu mask flips', int(fl.sum()))
