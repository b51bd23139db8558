import sys, torch
sys.path.insert(0,'/root/repo')
from qwen3_asr_mlx_amd import _lib, ops
dev=torch.device('cuda:0')
M,N,K=12480,4096,1024
A=torch.randn(M,K,device=dev).to(torch.bfloat16); W=torch.randn(N,K,device=dev).to(torch.bfloat16); out=torch.empty((M,N),dtype=torch.bfloat16,device=dev)
_lib.call("q3asr_gemm_force_kernel", 2)
ops.gemm(A,W,M,N,K,out); torch.cuda.synchronize()
