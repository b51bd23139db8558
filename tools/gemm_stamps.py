#!/usr/bin/env python3
"""Where a tile of the 8-phase GEMM spends its time: wall_clock64() stamps of workgroup 0 (q3asr_gemm_set_stamp)."""
import sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from qwen3_asr_mlx_amd import _lib, ops  # noqa: E402
dev = torch.device("cuda:0")
M = 12480
buf = torch.zeros(16, dtype=torch.int64, device=dev)
for (N, K, kind, act, name) in [(4096, 1024, "bf16", 0, "N=4096 K=1024 bf16"), (4096, 1024, "bf16", 1, "N=4096 K=1024 GELU (fc1)"),
                                (3072, 1024, "bf16", 0, "N=3072 K=1024 (enc qkv)"), (1024, 4096, "res32", 0, "N=1024 K=4096 fp32 residual (fc2)"),
                                (1024, 1024, "res32", 0, "N=1024 K=1024 fp32 residual (out)"), (4096, 2048, "bf16", 0, "N=4096 K=2048")]:
    A = torch.randn(M, K, device=dev).to(torch.bfloat16)
    W = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
    bias = torch.randn(N, device=dev) * 0.02
    out = torch.empty((M, N), dtype=torch.bfloat16, device=dev) if kind == "bf16" else torch.randn((M, N), dtype=torch.float32, device=dev)

    def call():
        if kind == "bf16":
            ops.gemm(A, W, M, N, K, out, bias=bias, act=act)
        else:
            ops.gemm(A, W, M, N, K, out, bias=bias, res=out, res_kind=1, ldres=N, out_fp32=True)
    for _ in range(3):
        call()
    rows = []
    for rep in range(5):
        buf.zero_()
        _lib.call("q3asr_gemm_set_stamp", ops.ptr(buf))
        call()
        torch.cuda.synchronize()
        _lib.call("q3asr_gemm_set_stamp", None)
        t = [v for v in buf.tolist() if v]
        rows.append([(b - a) / 100.0 for a, b in zip(t, t[1:])])          # 100 MHz -> us
    med = [sorted(r[i] for r in rows)[len(rows) // 2] for i in range(len(rows[0]))]
    labels = ["fill"] + [x for t in range(8) for x in (f"K{t}", f"epi{t}")]
    print(f"{name}: " + "  ".join(f"{l} {v:.1f}" for l, v in zip(labels, med)) + f"   | total {sum(med):.1f} us")

