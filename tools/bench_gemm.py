#!/usr/bin/env python3
"""Time q3asr_gemm_bf16 on the prefill / encoder shapes of the headline workload (random operands).
Usage: python tools/bench_gemm.py [rounds]; set Q3ASR_GEMM=128|256 to pin a kernel."""
import sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from qwen3_asr_mlx_amd import ops  # noqa: E402
dev = torch.device("cuda:0")
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
# (M, N, K, act, name): decoder prefill at 32 x 408 rows, encoder at 32 x 390 rows
shapes = [(13056, 12288, 2048, 2, "prefill gate/up+SwiGLU"), (13056, 4096, 2048, 0, "prefill qkv"), (13056, 2048, 2048, 0, "prefill o"),
          (13056, 2048, 6144, 0, "prefill down"), (12480, 4096, 1024, 1, "encoder fc1+GELU"), (12480, 1024, 4096, 0, "encoder fc2"),
          (12480, 1024, 1024, 0, "encoder q/k/v/out")]
ops_ = []
for (M, N, K, act, name) in shapes:
    A = torch.randn(M, K, device=dev).to(torch.bfloat16)
    W = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
    out = torch.empty((M, N // 2 if act == 2 else N), dtype=torch.bfloat16, device=dev)
    ops_.append((M, N, K, act, name, A, W, out))
best = {}
for r in range(rounds + 1):
    for (M, N, K, act, name, A, W, out) in ops_:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.gemm(A, W, M, N, K, out, act=act)
        e1.record()
        torch.cuda.synchronize()
        if r:
            best.setdefault(name, []).append(e0.elapsed_time(e1) / 10)
for (M, N, K, act, name, *_rest) in ops_:
    t = sorted(best[name])
    med = t[len(t) // 2]
    print(f"{name:26s} M={M:6d} N={N:6d} K={K:5d}  median {med * 1e3:8.1f} us  min {t[0] * 1e3:8.1f} us   {2 * M * N * K / med / 1e9:7.1f} TFLOP/s")
