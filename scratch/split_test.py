import sys; sys.path.insert(0,'/root/repo')
import torch
from glfusion_amd import ops
DEV='cuda'
def rel(a,t): return float((a.double().cpu()-t).abs().max()/t.abs().max())
def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/iters
torch.manual_seed(0)
for (M,N,K) in [(300,200,64),(1024,512,2048),(512,256,32768)]:
    A=torch.randn(M,K); B=torch.randn(N,K)
    ref=A.double()@B.double().T
    for mode in ('f32','bf16x6','f16x3'):
        ops.set_precision(mode)
        C=torch.empty(M,N,device=DEV)
        ops.gemm('nt',A.to(DEV),B.to(DEV),C,M=M,N=N,K=K,lda=K,ldb=K,ldc=N)
        print('nt',M,N,K,mode,'err',rel(C,ref))
    A2=torch.randn(K,M); B2=torch.randn(K,N); ref2=A2.double().T@B2.double()
    for mode in ('f32','bf16x6','f16x3'):
        ops.set_precision(mode)
        C=torch.zeros(M,N,device=DEV)
        ops.gemm('tn',A2.to(DEV),B2.to(DEV),C,M=M,N=N,K=K,lda=M,ldb=N,ldc=N,split=2)
        print('tn',M,N,K,mode,'err',rel(C,ref2))
for mode in ('f32','bf16x6','f16x3'):
    ops.set_precision(mode)
    M,N,K=150528,1024,2048
    A=torch.randn(M,K,device=DEV); B=torch.randn(N,K,device=DEV); C=torch.empty(M,N,device=DEV)
    ms=timeit(lambda: ops.gemm('nt',A,B,C,M=M,N=N,K=K,lda=K,ldb=K,ldc=N)); print(mode,f"nt {ms:.3f} ms {2*M*N*K/ms/1e9:.1f} TF (fp32-equivalent)")
    A2=torch.randn(150528,1024,device=DEV); B2=torch.randn(150528,2048,device=DEV); C2=torch.zeros(1024,2048,device=DEV)
    ms=timeit(lambda: ops.gemm('tn',A2,B2,C2,M=1024,N=2048,K=150528,lda=1024,ldb=2048,ldc=2048,split=16)); print(mode,f"tn {ms:.3f} ms {2*M*N*K/ms/1e9:.1f} TF")

# f16x3 with caller-supplied amax (no measuring pass) and range stress
ops.set_precision('f16x3')
from glfusion_amd._lib import lib
am=torch.zeros(2,device=DEV)
def amax_of(t,slot):
    lib.glf_amax(t.data_ptr(), 1, t.numel(), t.numel(), am[slot:].data_ptr(), None)
amax_of(A,0); amax_of(B,1)
ms=timeit(lambda: ops.gemm('nt',A,B,C,M=M,N=N,K=K,lda=K,ldb=K,ldc=N,amax_a=am[0:],amax_b=am[1:])); print(f"f16x3+amax nt {ms:.3f} ms {2*M*N*K/ms/1e9:.1f} TF")
amax_of(A2,0); amax_of(B2,1)
ms=timeit(lambda: ops.gemm('tn',A2,B2,C2,M=1024,N=2048,K=150528,lda=1024,ldb=2048,ldc=2048,split=16,amax_a=am[0:],amax_b=am[1:])); print(f"f16x3+amax tn {ms:.3f} ms {2*M*N*K/ms/1e9:.1f} TF")
ms=timeit(lambda: amax_of(A,0)); print(f"amax pass over {A.numel()*4/1e6:.0f} MB: {ms:.3f} ms")
for sa,sb in [(1e-12,1e9),(3e7,1e-3),(1e-30,1e20)]:
    M,N,K=512,256,1024
    a=(torch.randn(M,K)*sa); b=(torch.randn(N,K)*sb)
    a[5]*=1e-6   # a row far below the maximum
    ref=a.double()@b.double().T
    for mode in ('f32','bf16x6','f16x3'):
        ops.set_precision(mode)
        c=torch.empty(M,N,device=DEV)
        ops.gemm('nt',a.to(DEV),b.to(DEV),c,M=M,N=N,K=K,lda=K,ldb=K,ldc=N)
        print('range',sa,sb,mode,'err',rel(c,ref),'row5 rel',float((c[5].double().cpu()-ref[5]).abs().max()/ref[5].abs().max()))
