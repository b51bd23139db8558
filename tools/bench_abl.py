import sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from qwen3_asr_mlx_amd import ops  # noqa: E402
dev = torch.device("cuda:0")
B, MT = 32, 2
shapes = {"qkv": (4096, 2048, 4), "o": (2048, 2048, 8), "gu": (12288, 2048, 4), "down": (2048, 6144, 8)}
for name, (N, K, S) in shapes.items():
    Ws = [ops.to_w16((torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)) for _ in range(28)]
    x16 = torch.randn(MT * 16 * K, device=dev).to(torch.bfloat16)
    slab = torch.empty(16 * B * N, dtype=torch.float32, device=dev)
    for rep in range(3):
        for w in Ws:
            ops.skinny3(x16, B, w, N, K, 0, S, 4, slab=slab)
    torch.cuda.synchronize()
    del Ws
