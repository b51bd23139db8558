#!/usr/bin/env python3
"""A/B of the 8-phase GEMM's chained stream-K form in one process (q3asr_set_option(3, 0 | 1)): the encoder / prefill GEMM shapes at
batch 32, interleaved rounds, HIP events around 20 back-to-back launches per arm; operands rotated over 4 buffer sets.
Arm 0 = what rounds 3-4 shipped (whole tiles, 224- or 256-row, persistent where that was faster), arm 1 = stream-K (256-row tiles)."""
import sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from qwen3_asr_mlx_amd import _lib, ops  # noqa: E402
dev = torch.device("cuda:0")
lib = _lib.load()
shapes = [("enc fc1 (GELU)", 12480, 4096, 1024, dict(act=1, bias=True)), ("enc qkv", 12480, 3072, 1024, dict(bias=True)),
          ("enc fc2 (fp32 res)", 12480, 1024, 4096, dict(bias=True, res32=True)), ("enc out (fp32 res)", 12480, 1024, 1024, dict(bias=True, res32=True)),
          ("prefill gate/up (SwiGLU)", 13056, 12288, 2048, dict(act=2)), ("prefill down (bf16 res)", 13056, 2048, 6144, dict(res16=True)),
          ("prefill qkv", 13056, 4096, 2048, dict()), ("prefill o (bf16 res)", 13056, 2048, 2048, dict(res16=True))]
tot = {0: 0.0, 1: 0.0}
for name, M, N, K, kw in shapes:
    sets = []
    for i in range(4):
        A = torch.randn(M, K, device=dev).to(torch.bfloat16)
        W = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
        bias = torch.randn(N, device=dev) * 0.02 if kw.get("bias") else None
        if kw.get("res32"):
            out = torch.randn((M, N), dtype=torch.float32, device=dev)
        else:
            out = torch.empty((M, N // 2 if kw.get("act") == 2 else N), dtype=torch.bfloat16, device=dev)
        res = torch.randn((M, N), device=dev).to(torch.bfloat16) if kw.get("res16") else None
        sets.append((A, W, bias, out, res))

    def call(s):
        A, W, bias, out, res = s
        if kw.get("res32"):
            ops.gemm(A, W, M, N, K, out, bias=bias, res=out, res_kind=1, ldres=N, out_fp32=True)
        elif kw.get("res16"):
            ops.gemm(A, W, M, N, K, out, res=res, res_kind=2, ldres=N)
        else:
            ops.gemm(A, W, M, N, K, out, bias=bias, act=kw.get("act", 0))
    ts = {0: [], 1: []}
    outs = {}
    for rnd in range(6):
        for arm in (1, 0):
            lib.q3asr_set_option(3, arm)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(20):
                call(sets[i % 4])
            e1.record()
            torch.cuda.synchronize()
            if rnd:
                ts[arm].append(e0.elapsed_time(e1) * 1000 / 20)
            if not kw.get("res32"):
                outs[arm] = sets[3][3].clone()
    lib.q3asr_set_option(3, 1)
    same = "" if kw.get("res32") else f"  same bits: {bool(torch.equal(outs[0], outs[1]))}"
    med = lambda v: sorted(v)[len(v) // 2]      # noqa: E731
    tiles = ((M + 255) // 256) * ((N + 255) // 256)
    n_layers = 24 if name.startswith("enc") else 28
    tot[0] += n_layers * med(ts[0])
    tot[1] += n_layers * med(ts[1])
    tf = 2.0 * M * N * K / (med(ts[1]) * 1e-6) / 1e12
    print(f"{name:28s} {tiles:5d} tiles = {tiles / 256:5.2f} rounds   whole tiles {med(ts[0]):7.1f} us   stream-K {med(ts[1]):7.1f} us   "
          f"({100 * (med(ts[1]) / med(ts[0]) - 1):+.1f} %)  {tf:6.0f} TFLOP/s = {tf / 2500:.2f} of 2.5 PF{same}", flush=True)
print(f"per request (24 encoder + 28 decoder layers): whole tiles {tot[0] / 1e3:.2f} ms, stream-K {tot[1] / 1e3:.2f} ms")
