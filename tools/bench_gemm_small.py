#!/usr/bin/env python3
"""The GEMMs of ONE 30 s clip (encoder M = 390, prefill M = 408) and of small admission groups (k = 2, 4 clips): the 128x128-tile
kernel against the 64x64-tile kernel, interleaved in one process.  No split-K in either (one k-ascending chain per output)."""
import sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from qwen3_asr_mlx_amd import _lib, ops  # noqa: E402
dev = torch.device("cuda:0")
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
shapes = []
for k in (1, 2, 4):
    shapes += [(408 * k, 4096, 2048, 0, f"prefill qkv k={k}"), (408 * k, 2048, 2048, 0, f"prefill o k={k}"), (408 * k, 12288, 2048, 2, f"prefill gate/up k={k}"),
               (408 * k, 2048, 6144, 0, f"prefill down k={k}")]
for k in (1, 2, 4):
    shapes += [(390 * k, 4096, 1024, 1, f"encoder fc1 k={k}"), (390 * k, 1024, 4096, 0, f"encoder fc2 k={k}"), (390 * k, 3072, 1024, 0, f"encoder qkv k={k}"),
               (390 * k, 1024, 1024, 0, f"encoder out k={k}")]
ops_ = []
for (M, N, K, act, name) in shapes:
    A = torch.randn(M, K, device=dev).to(torch.bfloat16)
    W = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
    out = torch.empty((M, N // 2 if act == 2 else N), dtype=torch.bfloat16, device=dev)
    ops_.append((M, N, K, act, name, A, W, out))
best = {}
try:
    for r in range(rounds + 1):
        for (M, N, K, act, name, A, W, out) in ops_:
            for mode in (128, 64, 0):
                _lib.call("q3asr_gemm_force_kernel", mode)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    ops.gemm(A, W, M, N, K, out, act=act)
                e1.record()
                torch.cuda.synchronize()
                if r:
                    best.setdefault((name, mode), []).append(e0.elapsed_time(e1) / 20)
finally:
    _lib.call("q3asr_gemm_force_kernel", 0)
for (M, N, K, act, name, *_r) in ops_:
    t = {m: sorted(best[(name, m)])[len(best[(name, m)]) // 2] * 1e3 for m in (128, 64, 0)}
    print(f"{name:24s} M={M:5d} N={N:6d} K={K:5d}  128^2 {t[128]:7.1f} us   64^2 {t[64]:7.1f} us   auto {t[0]:7.1f} us", flush=True)
