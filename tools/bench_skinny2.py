import sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from qwen3_asr_mlx_amd import ops  # noqa: E402
dev = torch.device("cuda:0")
B, MT, L = 32, 2, 28
N, K = 12288, 2048
Ws = [ops.to_w16((torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)) for _ in range(L)]
x16 = torch.randn(MT * 16 * K, device=dev).to(torch.bfloat16)
out = torch.empty(B * N, dtype=torch.bfloat16, device=dev)
for rt in (2, 4):
    for waves in (2, 4):
        print(f"gu swiglu ksplit rt={rt} waves={waves}")
        for w in Ws:
            ops.skinny3(x16, B, w, N, K, 2, 1, waves, out=out, rt=rt, ksplit=True)
torch.cuda.synchronize()
del Ws
# lm_head
N, K = 151936, 2048
W = ops.to_w16((torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16))
flush = torch.empty(512 * 1024 * 1024 // 4, dtype=torch.float32, device=dev)
lg = torch.empty(B * N, dtype=torch.bfloat16, device=dev)
for rt in (1, 2, 4):
    for waves in (1, 2, 4):
        print(f"lm_head rt={rt} waves={waves}")
        for _ in range(6):
            flush.fill_(0.0)
            ops.skinny3(x16, B, W, N, K, 1, 1, waves, out=lg, rt=rt)
torch.cuda.synchronize()
