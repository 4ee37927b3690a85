import sys; sys.path.insert(0,'/root/repo')
import torch, copy
from oracle import glfusion_ref as orc
from glfusion_amd import ops
from glfusion_amd.models import Global_and_Local
DEV='cuda'
views,n,tv,start,temp=["1"],4,29,5,0.02
ref=orc.Global_and_Local(views); orc.closed_form_fill(ref,salt=5); orc.set_dropout(ref,0.0)
model=Global_and_Local(views); model.load_state_dict(ref.state_dict(),strict=True); orc.set_dropout(model,0.0); model=model.to(DEV).train(); ref.train()
imgs=orc.closed_form_images(views,n,112,112); tgts=orc.closed_form_targets(views,n)
video={v: orc.closed_form_tensor((tv,1,112,112),77,0.0,1.0) for v in views}
K='global_attn.g.weight'
def G(m): return dict(m.named_parameters())[K].grad.double().cpu().clone()
def zero(m):
    for p in m.parameters(): p.grad=None
res={}
for which in ('seg','cyc','both'):
    zero(ref); tot=0
    if which in('seg','both'):
        pred=ref(imgs)[0]; tot=tot+sum(torch.nn.functional.binary_cross_entropy_with_logits(pred[v],tgts[v],reduction='sum') for v in views)
    if which in('cyc','both'):
        feat=ref(video)[2]; tot=tot+1e-2*sum(orc.seg_cycle(feat[v].sum(dim=(2,3)),16,2,3,temp,start) for v in views)
    tot.backward(); r=G(ref)
    zero(model); tot=0
    if which in('seg','both'):
        pred=model({v:t.to(DEV) for v,t in imgs.items()})[0]; tot=tot+sum(ops.bce_with_logits_sum(pred[v],tgts[v].to(DEV)) for v in views)
    if which in('cyc','both'):
        feats=ops.pooled_fusion_features(model({v:t.to(DEV) for v,t in video.items()})[2]); tot=tot+1e-2*sum(ops.seg_cycle(feats[v],16,2,3,temp,start) for v in views)
    tot.backward(); e=G(model)
    print(which,'ref norm',float(r.norm()),'rel err',float((e-r).norm()/r.norm()))
