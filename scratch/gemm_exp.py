import sys, os; sys.path.insert(0, '/root/repo')
import glfusion_amd._lib as L
if len(sys.argv)>1: L.LIB_PATH=os.path.join(os.path.dirname(L.LIB_PATH), sys.argv[1])
import torch
from glfusion_amd import ops
ops.set_precision(sys.argv[2] if len(sys.argv)>2 else 'f32')
DEV='cuda'
def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/iters
M,N,K=150528,1024,2048
A=torch.randn(M,K,device=DEV); B=torch.randn(N,K,device=DEV); C=torch.empty(M,N,device=DEV)
ms=timeit(lambda: ops.gemm('nt',A,B,C,M=M,N=N,K=K,lda=K,ldb=K,ldc=N)); print(sys.argv[1:], f"nt {ms:.3f} ms {2*M*N*K/ms/1e9:.1f} TF")
