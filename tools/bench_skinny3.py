import sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from qwen3_asr_mlx_amd import ops  # noqa: E402
dev = torch.device("cuda:0")
B, MT, L = 32, 2, 28
shapes = {"qkv": (4096, 2048), "o": (2048, 2048), "down": (2048, 6144)}
for name, (N, K) in shapes.items():
    Ws = [ops.to_w16((torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)) for _ in range(L)]
    x16 = torch.randn(MT * 16 * K, device=dev).to(torch.bfloat16)
    slab = torch.empty(16 * B * N, dtype=torch.float32, device=dev)
    for rt in (1, 2):
        for waves in (2, 4):
            for S in (1, 2, 3, 4, 6):
                if K % (32 * S * waves) or (K // S) // waves < 64:
                    continue
                print(f"{name} ksplit rt={rt} waves={waves} S={S}")
                for w in Ws:
                    ops.skinny3(x16, B, w, N, K, 0, S, waves, slab=slab, rt=rt, ksplit=True)
    torch.cuda.synchronize()
    del Ws
