import sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from qwen3_asr_mlx_amd import ops  # noqa: E402
dev = torch.device("cuda:0")
shapes = [(13056, 12288, 2048, 2), (13056, 4096, 2048, 0), (13056, 2048, 6144, 0), (12480, 4096, 1024, 1), (12480, 1024, 4096, 0)]
for (M, N, K, act) in shapes:
    A = (torch.randn(M, K, device=dev)).to(torch.bfloat16)
    W = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
    out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    for _ in range(5):
        ops.gemm(A, W, M, N, K, out, act=act)
    torch.cuda.synchronize()
