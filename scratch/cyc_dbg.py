import sys; sys.path.insert(0,'/root/repo')
import torch, copy
from oracle import glfusion_ref as orc
from glfusion_amd import ops
from glfusion_amd.models import Global_and_Local
DEV='cuda'
views,n,tv,start=["1"],4,29,5
ref=orc.Global_and_Local(views); orc.closed_form_fill(ref,salt=5); orc.set_dropout(ref,0.0)
model=Global_and_Local(views); model.load_state_dict(ref.state_dict(),strict=True); orc.set_dropout(model,0.0); model=model.to(DEV).train(); ref.train()
ref64=copy.deepcopy(ref).double()
video={v: orc.closed_form_tensor((tv,1,112,112),77,0.0,1.0) for v in views}
def cyc_grads(m, vid, dt):
    for p in m.parameters(): p.grad=None
    feat=m({v:t.to(dt) for v,t in vid.items()})[2]
    c=sum(orc.seg_cycle(feat[v].sum(dim=(2,3)),16,2,3,10,start) for v in views)
    c.backward()
    return float(c), {k:p.grad.double().clone() for k,p in m.named_parameters() if p.grad is not None and k.startswith('global_attn') }, feat
c32,g32,f32=cyc_grads(ref,video,torch.float32)
c64,g64,f64=cyc_grads(ref64,video,torch.float64)
for p in model.parameters(): p.grad=None
fe=model({v:t.to(DEV) for v,t in video.items()})[2]
feats=ops.pooled_fusion_features(fe)
ce=sum(ops.seg_cycle(feats[v],16,2,3,10,start) for v in views); ce.backward()
ge={k:p.grad.double().cpu() for k,p in model.named_parameters() if p.grad is not None and k.startswith('global_attn')}
print('cyc loss: ref32',c32,'ref64',c64,'engine',float(ce))
pf32=f32['1'].sum(dim=(2,3)).double(); pf64=f64['1'].sum(dim=(2,3)); pfe=feats['1'].double().cpu()
print('pooled feat err: ref32',float((pf32-pf64).abs().max()/pf64.abs().max()),'engine',float((pfe-pf64).abs().max()/pf64.abs().max()), 'max',float(pf64.abs().max()))
for k in list(g64)[:8]:
    w=g64[k]; print(k, 'ref32 rel', float((g32[k]-w).norm()/w.norm()), 'engine rel', float((ge[k]-w).norm()/w.norm()))
