import sys; sys.path.insert(0, '/root/repo')
import torch, numpy as np
import torch.nn.functional as F
from oracle import glfusion_ref as orc
from glfusion_amd import ops
from glfusion_amd.models import resnet as hip_resnet
DEV='cuda'
def err(a,b):
    a=a.detach().cpu().double(); b=b.detach().cpu().double()
    return float((a-b).abs().max()), float(b.abs().max())
torch.manual_seed(0)
for cfg in [(2,12,12,512,512,3,1,4,4),(2,12,12,512,512,3,1,2,2),(2,12,12,1024,2048,1,1,0,1),(2,12,12,2048,512,1,1,0,1),(2,28,28,2048,256,3,1,12,12)]:
    n,h,w,cin,cout,k,s,p,d=cfg
    x=torch.rand(n,cin,h,w); wt=(torch.rand(cout,cin,k,k)-0.5)/np.sqrt(cin*k*k)
    ref=F.conv2d(x.double(),wt.double(),None,s,p,d)
    ref32=F.conv2d(x,wt,None,s,p,d)
    y=ops.conv2d(x.permute(0,2,3,1).contiguous().to(DEV), wt.to(DEV), None, s,p,d).permute(0,3,1,2)
    print(cfg, 'hip vs f64', err(y,ref), 'cpu32 vs f64', err(ref32,ref))
trunk_o = orc.ResNet50Trunk((False, True, True)); trunk_h = hip_resnet.ResNet((3,4,6,3),(False,True,True))
orc.closed_form_fill(trunk_o, salt=9); trunk_h.load_state_dict(trunk_o.state_dict()); trunk_h=trunk_h.to(DEV)
lo, lh = trunk_o.layer4.train(), trunk_h.layer4.train()
x = orc.closed_form_tensor((2,1024,12,12),300,0.0,1.0)
xo=x; xh=ops.to_nhwc(x.to(DEV))
for bi,(bo,bh) in enumerate(zip(lo,lh)):
    # step through ops
    co1=bo.conv1(xo); ch1=bh.conv1.forward_nhwc(xh); print(bi,'conv1',err(ch1.permute(0,3,1,2),co1))
    o1=bo.relu(bo.bn1(co1)); h1=bh.bn1.forward_nhwc(ch1,relu=True); print(bi,'bn1',err(h1.permute(0,3,1,2),o1))
    co2=bo.conv2(o1); ch2=bh.conv2.forward_nhwc(h1); print(bi,'conv2',err(ch2.permute(0,3,1,2),co2))
    o2=bo.relu(bo.bn2(co2)); h2=bh.bn2.forward_nhwc(ch2,relu=True); print(bi,'bn2',err(h2.permute(0,3,1,2),o2))
    co3=bo.conv3(o2); ch3=bh.conv3.forward_nhwc(h2); print(bi,'conv3',err(ch3.permute(0,3,1,2),co3))
    if bo.downsample is not None:
        so=bo.downsample(xo); sh=bh.downsample[1].forward_nhwc(bh.downsample[0].forward_nhwc(xh)); print(bi,'down',err(sh.permute(0,3,1,2),so))
    else: so,sh=xo,xh
    xo=bo.relu(bo.bn3(co3)+so); xh=bh.bn3.forward_nhwc(ch3,relu=True,residual=sh); print(bi,'out',err(xh.permute(0,3,1,2),xo))
    # continue from the oracle's value to localise
    xh=ops.to_nhwc(xo.to(DEV))
