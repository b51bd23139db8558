#!/usr/bin/env python3
"""Square / long-K GEMM timings for q3asr_gemm_bf16 (random operands): where the kernels sit against the 2.5 PFLOP/s peak."""
import sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from qwen3_asr_mlx_amd import ops  # noqa: E402
dev = torch.device("cuda:0")
for (M, N, K) in [(4096, 4096, 4096), (8192, 8192, 8192), (4096, 4096, 16384), (16384, 16384, 2048)]:
    A = (torch.rand(M, K, device=dev) * 2 - 1).to(torch.bfloat16)
    W = (torch.rand(N, K, device=dev) * 2 - 1).to(torch.bfloat16)
    out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    ts = []
    for r in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            ops.gemm(A, W, M, N, K, out)
        e1.record()
        torch.cuda.synchronize()
        if r:
            ts.append(e0.elapsed_time(e1) / 5)
    ts.sort()
    print(f"M={M} N={N} K={K}: median {ts[len(ts)//2]*1e3:9.1f} us  {2*M*N*K/ts[len(ts)//2]/1e9:7.1f} TFLOP/s")
