import sys; sys.path.insert(0, '/root/repo')
import torch, time
from glfusion_amd import ops
DEV='cuda'
def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/iters
def plain(mode, M,N,K, batch=1):
    if mode=='nt':
        A=torch.randn(batch,M,K,device=DEV); B=torch.randn(batch,N,K,device=DEV); C=torch.empty(batch,M,N,device=DEV)
        f=lambda: ops.gemm('nt',A,B,C,M=M,N=N,K=K,lda=K,ldb=K,ldc=N,batch=batch,bsa=M*K,bsb=N*K,bsc=M*N)
    elif mode=='nn':
        A=torch.randn(batch,M,K,device=DEV); B=torch.randn(batch,K,N,device=DEV); C=torch.empty(batch,M,N,device=DEV)
        f=lambda: ops.gemm('nn',A,B,C,M=M,N=N,K=K,lda=K,ldb=N,ldc=N,batch=batch,bsa=M*K,bsb=N*K,bsc=M*N)
    else:  # tn: K = rows
        A=torch.randn(batch,K,M,device=DEV); B=torch.randn(batch,K,N,device=DEV); C=torch.zeros(batch,M,N,device=DEV)
        sp=ops._tn_split(K,M,N,1,batch)
        f=lambda: ops.gemm('tn',A,B,C,M=M,N=N,K=K,lda=M,ldb=N,ldc=N,batch=batch,bsa=K*M,bsb=K*N,bsc=M*N,split=sp)
    ms=timeit(f); print(f"{mode} M={M} N={N} K={K} b={batch}: {ms:.3f} ms  {2*M*N*K*batch/ms/1e9:.1f} TF")
def conv(n,h,cin,cout,k,dil):
    x=torch.randn(n,h,h,cin,device=DEV); w=torch.randn(cout,cin,k,k,device=DEV)
    f=lambda: ops.conv2d(x,w,None,1,dil if k==3 else 0,dil)
    ms=timeit(f); fl=2*n*h*h*cin*cout*k*k
    print(f"conv fwd n={n} {h}x{h} {cin}->{cout} k{k} d{dil}: {ms:.3f} ms dense {fl/ms/1e9:.1f} TF")
plain('nt',150528,1024,2048)
plain('nt',50176,2048,512)
plain('nt',50176,256,2048)
plain('nn',150528,2048,1024)
plain('tn',1024,2048,150528)
plain('tn',2048,512,50176)
plain('nt',2352,1024,1024,64)
plain('tn',1024,1024,2352,64)
conv(64,28,2048,256,3,12)
conv(64,28,2048,256,3,24)
conv(64,28,512,512,3,4)
conv(64,28,256,256,3,2)
conv(64,55,64,64,3,1)
