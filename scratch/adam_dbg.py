import sys; sys.path.insert(0,'/root/repo')
import torch, copy
from glfusion_amd.optim import Adam
DEV='cuda'
p0=torch.randn(7)*0.3
g1=torch.randn(7); g2=torch.randn(7)
a=torch.nn.Parameter(p0.clone().to(DEV)); o=Adam([a],lr=1e-3,weight_decay=1e-5)
a.grad=g1.to(DEV); o.step()
sd=copy.deepcopy(o.state_dict()); print('ours sd step', sd['state'][0]['step'])
t=torch.nn.Parameter(a.detach().clone()); to=torch.optim.Adam([t],lr=1e-3,weight_decay=1e-5); to.load_state_dict(sd)
print('torch state step after load', to.state[t]['step'])
b=Adam([a],lr=1e-3,weight_decay=1e-5); b.load_state_dict(to.state_dict()); print('back step', b.state[a]['step'])
a.grad=g2.to(DEV); t.grad=g2.to(DEV)
b.step(); to.step()
print('steps after', b.state[a]['step'], to.state[t]['step'])
print((a-t).abs().max().item())
# cpu reference chain
c=torch.nn.Parameter(p0.clone()); co=torch.optim.Adam([c],lr=1e-3,weight_decay=1e-5)
c.grad=g1.clone(); co.step(); c.grad=g2.clone(); co.step()
print('vs cpu: ours',(a.detach().cpu()-c).abs().max().item(),'torch gpu',(t.detach().cpu()-c).abs().max().item())
