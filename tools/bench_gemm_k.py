#!/usr/bin/env python3
"""K sweep of q3asr_gemm_bf16 at the encoder's M (12,480 rows): separates the per-tile fixed cost (pipeline fill, epilogue,
round quantisation) from the main-loop rate.  Usage: python3 tools/bench_gemm_k.py [rounds]
"""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from qwen3_asr_mlx_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
M = 12480
cases = []
for N in (4096, 3072, 1024):
    for K in (256, 512, 1024, 2048, 4096):
        cases.append((N, K, "bf16", 0))
for N, K in ((1024, 1024), (1024, 4096)):
    cases.append((N, K, "res32", 0))
cases.append((4096, 1024, "bf16", 1))          # fc1 + GELU
bufs = {}
for (N, K, kind, act) in cases:
    A = torch.randn(M, K, device=dev).to(torch.bfloat16)
    W = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
    bias = torch.randn(N, device=dev) * 0.02
    if kind == "bf16":
        out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
        call = (lambda A=A, W=W, N=N, K=K, out=out, bias=bias, act=act: ops.gemm(A, W, M, N, K, out, bias=bias, act=act))
    else:
        out = torch.randn((M, N), dtype=torch.float32, device=dev)
        call = (lambda A=A, W=W, N=N, K=K, out=out, bias=bias: ops.gemm(A, W, M, N, K, out, bias=bias, res=out, res_kind=1, ldres=N, out_fp32=True))
    bufs[(N, K, kind, act)] = call
best = {}
for r in range(rounds + 1):
    for key, call in bufs.items():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            call()
        e1.record()
        torch.cuda.synchronize()
        if r:
            best.setdefault(key, []).append(e0.elapsed_time(e1) / 10)
for (N, K, kind, act) in cases:
    t = sorted(best[(N, K, kind, act)])
    med = t[len(t) // 2]
    tiles = ((M + 255) // 256) * ((N + 255) // 256)
    per_cu = -(-tiles // 256)
    print(f"N={N:5d} K={K:5d} {kind:5s} act={act}  {med * 1e3:8.1f} us  {2 * M * N * K / med / 1e9:7.1f} TFLOP/s  tiles {tiles:4d} "
          f"(max {per_cu}/CU -> {med * 1e3 / per_cu:6.1f} us per tile slot)")
