import sys; sys.path.insert(0,'/root/repo')
import torch
from glfusion_amd import ops
from glfusion_amd._lib import lib
DEV='cuda'
def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/iters
ops.set_precision('f16x3')
am=torch.ones(2,device=DEV)*4
for (M,N,K) in [(150528,1024,2048),(150528,512,512),(37632,2048,1024)]:
    A=torch.randn(M,K,device=DEV); B=torch.randn(N,K,device=DEV); C=torch.empty(M,N,device=DEV)
    ms=timeit(lambda: ops.gemm('nt',A,B,C,M=M,N=N,K=K,lda=K,ldb=K,ldc=N,amax_a=am[0:],amax_b=am[1:])); print(f"nt {M}x{N}x{K} {ms:.3f} ms {2*M*N*K/ms/1e9:.1f} TF")
