#!/usr/bin/env python3
"""Sweep of the decode weight-streaming GEMM (skinny3) on cold weights: 28 distinct weight matrices per
shape (footprint > 256 MiB Infinity Cache).  Run under `rocprofv3 --kernel-trace` and read the per-config
kernel durations with tools/agg_trace.py (configs are told apart by grid x, grid y, workgroup size)."""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from qwen3_asr_mlx_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
MT = ops.mt_of(B)
L = 28
shapes = {"qkv": (4096, 2048), "o": (2048, 2048), "gu": (12288, 2048), "down": (2048, 6144)}
for name, (N, K) in shapes.items():
    Ws = [ops.to_w16((torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)) for _ in range(L)]
    x16 = torch.randn(MT * 16 * K, device=dev).to(torch.bfloat16)
    slab = torch.empty(16 * B * N, dtype=torch.float32, device=dev)
    out = torch.empty(B * N, dtype=torch.bfloat16, device=dev)
    for rt in (1, 2, 4):
        for waves in (1, 2, 4):
            for S in (1, 2, 3, 4, 6, 8, 12):
                if K % (32 * S) or K // S < 128:
                    continue
                tiles = (N // 16 + waves * rt - 1) // (waves * rt)
                if tiles * S < 128 or tiles * S > 2048:
                    continue
                print(f"{name} N={N} K={K} rt={rt} waves={waves} S={S}: grid=({tiles*64*waves},{S}) wg={64*waves}")
                for w in Ws:
                    ops.skinny3(x16, B, w, N, K, 0, S, waves, slab=slab, rt=rt)
    if name == "gu":
        for rt in (2, 4):
            for waves in (1, 2, 4):
                tiles = (N // 16) // (waves * rt)
                print(f"gu swiglu rt={rt} waves={waves}: grid=({tiles*64*waves},1) wg={64*waves} EPI2")
                for w in Ws:
                    ops.skinny3(x16, B, w, N, K, 2, 1, waves, out=out, rt=rt)
    torch.cuda.synchronize()
    del Ws
