import sys, os; sys.path.insert(0,'/root/repo')
import torch
from glfusion_amd import ops
ops.set_precision('bf16x6')
DEV='cuda'
def timeit(fn, iters=20):
    fn(); torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/iters
def plain(M,N,K):
    A=torch.randn(M,K,device=DEV); B=torch.randn(N,K,device=DEV); C=torch.empty(M,N,device=DEV)
    ms=timeit(lambda: ops.gemm('nt',A,B,C,M=M,N=N,K=K,lda=K,ldb=K,ldc=N)); return ms, 2*M*N*K/ms/1e9
def conv(n,h,cin,cout,k,dil):
    x=torch.randn(n,h,h,cin,device=DEV); w=torch.randn(cout,cin,k,k,device=DEV)
    ms=timeit(lambda: ops.conv2d(x,w,None,1,dil if k==3 else 0,dil)); return ms, 2*n*h*h*cin*cout*k*k/ms/1e9
tag=os.environ.get('GLF_BF16S_V1','v2')
for s in [(50176,256,256),(50176,1024,256),(50176,256,1024),(50176,512,2048),(50176,2048,512),(193600,256,64),(193600,64,256),(50176,512,128),(50176,128,512),(2352*8,1024,1024)]:
    ms,tf=plain(*s); print(tag,'nt',s,f"{ms:.3f} ms {tf:.0f} TF")
for c in [(64,28,256,256,3,2),(64,28,128,128,3,1),(64,55,64,64,3,1),(64,28,512,512,3,4)]:
    ms,tf=conv(*c); print(tag,'conv',c,f"{ms:.3f} ms {tf:.0f} TF")
