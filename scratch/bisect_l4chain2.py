import sys, os; sys.path.insert(0, '/root/repo')
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
xd=x.double().requires_grad_(True); yd=trunk_d.layer4.train()(xd)
gy=orc.closed_form_tensor(tuple(yd.shape),301,-1.0,1.0)
yd.backward(gy.double())
L=trunk_h.layer4.train()
def run(mode):
    xh=x.to(DEV).requires_grad_(True)
    a=xh
    for blk in L:
        if mode=='clone': a=blk(a).contiguous().clone()
        elif mode=='nhwc_clone': a=blk(a); a=ops.from_nhwc(ops.to_nhwc(a).clone())
        else: a=blk(a)
    a.backward(gy.to(DEV))
    torch.cuda.synchronize()
    return rel(xh.grad, xd.grad), rel(a, yd)
for mode in ('plain','plain','clone','nhwc_clone'):
    print(mode, run(mode))
# last two blocks only, chained, from oracle input
