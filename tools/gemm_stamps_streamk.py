#!/usr/bin/env python3
"""Chained stream-K against whole tiles, one workgroup's timeline: wall_clock64() stamps (q3asr_gemm_set_stamp / _wg) of workgroup
blockIdx 8 (stream-K: range index 1 -- it has a head piece, whole slots and a tail piece) for the encoder's fc1 and the prefill's o."""
import sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from qwen3_asr_mlx_amd import _lib, ops  # noqa: E402
dev = torch.device("cuda:0")
lib = _lib.load()
buf = torch.zeros(16, dtype=torch.int64, device=dev)
for (M, N, K, act, name) in [(12480, 4096, 1024, 1, "enc fc1 (GELU)"), (12480, 3072, 1024, 0, "enc qkv"), (13056, 2048, 2048, 0, "prefill o (no residual here)"),
                             (13056, 4096, 2048, 0, "prefill qkv")]:
    A = torch.randn(M, K, device=dev).to(torch.bfloat16)
    W = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
    bias = torch.randn(N, device=dev) * 0.02
    out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    for arm, mode in ((0, 8), (0, 0), (1, 8)):
        lib.q3asr_set_option(3, arm)
        _lib.call("q3asr_gemm_force_kernel", mode)
        for _ in range(3):
            ops.gemm(A, W, M, N, K, out, bias=bias, act=act)
        rows, tot = [], []
        for rep in range(7):
            buf.zero_()
            lib.q3asr_gemm_set_stamp_wg(8)
            _lib.call("q3asr_gemm_set_stamp", ops.ptr(buf))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ops.gemm(A, W, M, N, K, out, bias=bias, act=act)
            e1.record()
            torch.cuda.synchronize()
            _lib.call("q3asr_gemm_set_stamp", None)
            t = [v for v in buf.tolist() if v]
            rows.append([(b - a) / 100.0 for a, b in zip(t, t[1:])])          # 100 MHz -> us
            tot.append(e0.elapsed_time(e1) * 1e3)
        n = min(len(r) for r in rows)
        med = [sorted(r[i] for r in rows)[len(rows) // 2] for i in range(n)]
        labels = ["fill"] + [x for t in range(8) for x in (f"K{t}", f"end{t}")]
        what = {(0, 8): "whole 256-row tiles", (0, 0): "whole tiles, auto height", (1, 8): "chained stream-K"}[(arm, mode)]
        print(f"{name:30s} {what:26s}: " + "  ".join(f"{l} {v:.1f}" for l, v in zip(labels, med)) + f"   | wg sum {sum(med):.1f} us, launch (events) {sorted(tot)[len(tot) // 2]:.1f} us", flush=True)
lib.q3asr_set_option(3, 1)
_lib.call("q3asr_gemm_force_kernel", 0)
