import sys; sys.path.insert(0, '/root/repo')
import torch
from glfusion_amd import ops
DEV='cuda'
M,N,K=150528,1024,2048
A=torch.randn(M,K,device=DEV); B=torch.randn(N,K,device=DEV); C=torch.empty(M,N,device=DEV)
for _ in range(3): ops.gemm('nt',A,B,C,M=M,N=N,K=K,lda=K,ldb=K,ldc=N)
A2=torch.randn(K and 150528, 1024, device=DEV); B2=torch.randn(150528,2048,device=DEV); C2=torch.zeros(1024,2048,device=DEV)
for _ in range(3): ops.gemm('tn',A2,B2,C2,M=1024,N=2048,K=150528,lda=1024,ldb=2048,ldc=2048,split=16)
torch.cuda.synchronize()
