#!/usr/bin/env python3
"""A/B of the 8-phase GEMM's interior-tile epilogue (and, round 5, of its residual-ahead form) in one process (q3asr_gemm_set_interior 0 / 1 / 2): the encoder / prefill GEMM shapes
at batch 32, interleaved rounds, HIP events around 20 back-to-back launches per arm; operands rotated over 4 buffer sets."""
import ctypes as C
import sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from qwen3_asr_mlx_amd import _lib, ops  # noqa: E402
dev = torch.device("cuda:0")
lib = _lib.load()
lib.q3asr_gemm_set_interior.argtypes = [C.c_int]
shapes = [("enc fc1 (GELU)", 12480, 4096, 1024, dict(act=1, bias=True)), ("enc qkv", 12480, 3072, 1024, dict(bias=True)),
          ("enc fc2 (fp32 res)", 12480, 1024, 4096, dict(bias=True, res32=True)), ("enc out (fp32 res)", 12480, 1024, 1024, dict(bias=True, res32=True)),
          ("prefill gate/up (SwiGLU)", 13056, 12288, 2048, dict(act=2)), ("prefill down (bf16 res)", 13056, 2048, 6144, dict(res16=True)),
          ("prefill qkv", 13056, 4096, 2048, dict()), ("prefill o (bf16 res)", 13056, 2048, 2048, dict(res16=True))]
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
    ts = {0: [], 1: [], 2: []}
    outs = {}
    for rnd in range(6):
        for arm in (2, 1, 0):
            lib.q3asr_gemm_set_interior(arm)
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
    lib.q3asr_gemm_set_interior(2)
    same = "" if kw.get("res32") else f"  same bits: {bool(torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]))}"
    med = lambda v: sorted(v)[len(v) // 2]      # noqa: E731
    print(f"{name:28s} general {med(ts[0]):7.1f} us   interior {med(ts[1]):7.1f} us   ({100 * (med(ts[1]) / med(ts[0]) - 1):+.1f} %)   "
          f"+ residual rows one block ahead {med(ts[2]):7.1f} us   ({100 * (med(ts[2]) / med(ts[1]) - 1):+.1f} % vs interior){same}")
