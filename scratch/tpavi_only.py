"""The fusion block alone at the C2 shape (one TPAVIModule, [64, 3, 28, 28, 2048], train fwd + bwd), for rocprofv3."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from glfusion_amd import ops
from glfusion_amd.models import TPAVIModule
prec = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
ops.set_precision(prec)
torch.manual_seed(0)
m = TPAVIModule(in_channels=2048, mode="dot").cuda().train()
with torch.no_grad():
    m.W_z[1].weight.normal_(1.0, 0.1)
x = torch.randn(64, 3, 28, 28, 2048, device="cuda", requires_grad=True)
w = torch.randn_like(x)
for it in range(3):
    z = m.forward_nvhwc(x)
    z.backward(w)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for it in range(3):
    z = m.forward_nvhwc(x)
    z.backward(w)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 3
# dense MACs per frame of one module: projections 19.73 G, re-associated attention 4.93 G; x3 for fwd+bwd
gmac = (19.73 + 4.93) * 64 * 3
print(f"{prec}: TPAVI fwd+bwd {ms:.2f} ms, {2 * gmac / ms:.1f} TFLOP/s fp32-equivalent (executed, re-associated)")
