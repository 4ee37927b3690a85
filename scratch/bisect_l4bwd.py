import sys; sys.path.insert(0, '/root/repo')
import torch, numpy as np, copy
import torch.nn.functional as F
from oracle import glfusion_ref as orc
from glfusion_amd import ops
from glfusion_amd.models import resnet as hip_resnet
DEV='cuda'
def rel(a,t):
    a=a.detach().cpu().double(); t=t.detach().cpu().double()
    return float((a-t).abs().max())/max(float(t.abs().max()),1e-30)
torch.manual_seed(0)
for cfg in [(2,12,12,512,512,3,1,4,4),(2,12,12,512,512,3,1,2,2),(2,12,12,1024,2048,1,1,0,1),(2,12,12,2048,512,1,1,0,1),(2,12,12,512,2048,1,1,0,1)]:
    n,h,w,cin,cout,k,s,p,d=cfg
    x=torch.rand(n,cin,h,w); wt=(torch.rand(cout,cin,k,k)-0.5)/np.sqrt(cin*k*k)
    outs={}
    for tag in ('d','c','h'):
        xx = x.double() if tag=='d' else x.clone(); ww = wt.double() if tag=='d' else wt.clone()
        if tag=='h':
            xx=xx.permute(0,2,3,1).contiguous().to(DEV).requires_grad_(True); ww=ww.to(DEV).requires_grad_(True)
            y=ops.conv2d(xx,ww,None,s,p,d)
            gy=torch.rand(*y.shape, generator=torch.Generator().manual_seed(5))  # NHWC
            y.backward(gy.to(DEV)); outs[tag]=(y.permute(0,3,1,2), xx.grad.permute(0,3,1,2), ww.grad)
        else:
            xx.requires_grad_(True); ww.requires_grad_(True)
            y=F.conv2d(xx,ww,None,s,p,d)
            gy=torch.rand(y.shape[0],y.shape[2],y.shape[3],y.shape[1], generator=torch.Generator().manual_seed(5)).permute(0,3,1,2)
            y.backward(gy.to(y.dtype)); outs[tag]=(y,xx.grad,ww.grad)
    print(cfg,'y %.1e/%.1e dx %.1e/%.1e dw %.1e/%.1e'%(rel(outs['h'][0],outs['d'][0]),rel(outs['c'][0],outs['d'][0]),rel(outs['h'][1],outs['d'][1]),rel(outs['c'][1],outs['d'][1]),rel(outs['h'][2],outs['d'][2]),rel(outs['c'][2],outs['d'][2])))
# per-block
trunk_o = orc.ResNet50Trunk((False, True, True)); orc.closed_form_fill(trunk_o, salt=9)
trunk_h = hip_resnet.ResNet((3,4,6,3),(False,True,True)); trunk_h.load_state_dict(trunk_o.state_dict()); trunk_h=trunk_h.to(DEV)
trunk_d = copy.deepcopy(trunk_o).double()
x = orc.closed_form_tensor((2,1024,12,12),300,0.0,1.0)
xo=x
for bi in range(3):
    bo,bh,bd = trunk_o.layer4[bi].train(), trunk_h.layer4[bi].train(), trunk_d.layer4[bi].train()
    a=xo.clone().requires_grad_(True); b=xo.double().requires_grad_(True); c=xo.to(DEV).requires_grad_(True)
    yo,yd,yh=bo(a),bd(b),bh(c)
    gy=orc.closed_form_tensor(tuple(yo.shape),301,-1.0,1.0)
    yo.backward(gy); yd.backward(gy.double()); yh.backward(gy.to(DEV))
    print('block',bi,'y %.1e/%.1e dx %.1e/%.1e'%(rel(yh,yd),rel(yo,yd),rel(c.grad,b.grad),rel(a.grad,b.grad)))
    for (n1,p1),(n2,p2),(n3,p3) in zip(bo.named_parameters(),bh.named_parameters(),bd.named_parameters()):
        print('    ',n1,'%.1e/%.1e'%(rel(p2.grad,p3.grad),rel(p1.grad,p3.grad)))
    xo=yo.detach()
