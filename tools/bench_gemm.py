#!/usr/bin/env python3
"""Time q3asr_gemm_bf16 on the prefill / encoder shapes of the headline workload (random operands), kernels interleaved in
one process.  Usage: python tools/bench_gemm.py [rounds] [modes=0]  (modes: comma list of 0 auto, 64, 128, 8 = 8-phase 256-row tiles, 224 = 8-phase 224-row tiles)."""
import sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from qwen3_asr_mlx_amd import _lib, ops  # noqa: E402
dev = torch.device("cuda:0")
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
modes = [int(m) for m in (sys.argv[2] if len(sys.argv) > 2 else "0").split(",")]
# (M, N, K, act, name): decoder prefill at 32 x 408 rows, encoder at 32 x 390 rows
# (M, N, K, act, residual kind, fp32 out, name): the call sites' own epilogues
shapes = [(13056, 12288, 2048, 2, 0, False, "prefill gate/up+SwiGLU"), (13056, 4096, 2048, 0, 0, False, "prefill qkv"),
          (13056, 2048, 2048, 0, 2, False, "prefill o +res"), (13056, 2048, 6144, 0, 2, False, "prefill down +res"),
          (12480, 4096, 1024, 1, 0, False, "encoder fc1+GELU"), (12480, 1024, 4096, 0, 1, True, "encoder fc2 +res f32"),
          (12480, 3072, 1024, 0, 0, False, "encoder qkv"), (12480, 1024, 1024, 0, 1, True, "encoder out +res f32")]
ops_ = []
for (M, N, K, act, rk, f32, name) in shapes:
    A = torch.randn(M, K, device=dev).to(torch.bfloat16)
    W = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
    out = torch.empty((M, N // 2 if act == 2 else N), dtype=torch.float32 if f32 else torch.bfloat16, device=dev)
    res = torch.randn(M, N, device=dev).to(torch.float32 if rk == 1 else torch.bfloat16) if rk else None
    bias = torch.randn(N, device=dev) if act == 1 or rk == 1 else None
    ops_.append((M, N, K, act, rk, f32, name, A, W, out, res, bias))
best = {}
try:
    for r in range(rounds + 1):
        for (M, N, K, act, rk, f32, name, A, W, out, res, bias) in ops_:
            for mode in modes:
                _lib.call("q3asr_gemm_force_kernel", mode)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    ops.gemm(A, W, M, N, K, out, act=act, bias=bias, res=res, res_kind=rk, ldres=N if rk else 0, out_fp32=f32)
                e1.record()
                torch.cuda.synchronize()
                if r:
                    best.setdefault((name, mode), []).append(e0.elapsed_time(e1) / 10)
finally:
    _lib.call("q3asr_gemm_force_kernel", 0)
for (M, N, K, act, rk, f32, name, *_rest) in ops_:
    for mode in modes:
        t = sorted(best[(name, mode)])
        med = t[len(t) // 2]
        print(f"{name:24s} mode {mode:3d} M={M:6d} N={N:6d} K={K:5d}  median {med * 1e3:8.1f} us  min {t[0] * 1e3:8.1f} us   {2 * M * N * K / med / 1e9:7.1f} TFLOP/s", flush=True)
