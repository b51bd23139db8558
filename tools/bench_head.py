#!/usr/bin/env python3
"""The tied head of the decode step (151,936 x 2,048 bf16 = 622 MB streamed per token) on skinny4, epilogue 1: rows 1 / 16 / 32 and
the instantiated geometries, three weight copies rotated so that no launch finds its weights in the Infinity Cache."""
import sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from qwen3_asr_mlx_amd import ops  # noqa: E402
dev = torch.device("cuda:0")
N, K = 151936, 2048
geos = [tuple(int(v) for v in g.split(",")) for g in (sys.argv[1:] or ["2,8,8", "4,8,8"])]
Ws = [ops.to_w16((torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)) for _ in range(3)]
for M in (1, 8, 16, 17, 24, 32):
    MT = ops.mt_of(M)
    x16 = ops.to_x16(torch.randn(M, K, device=dev).to(torch.bfloat16), MT)
    out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    ref = None
    for g in geos:
        tpw, nks, waves = g
        ts = []
        for rnd in range(6):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(12):
                ops.skinny4(x16, M, Ws[i % 3], N, K, 1, tpw, nks, waves, out=out)
            e1.record()
            torch.cuda.synchronize()
            if rnd:
                ts.append(e0.elapsed_time(e1) * 1000 / 12)
        from qwen3_asr_mlx_amd import _lib
        _lib.load().q3asr_set_option(2, 0)                                        # the per-group launch form of the same geometry
        tt = []
        for rnd in range(6):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(12):
                ops.skinny4(x16, M, Ws[i % 3], N, K, 1, tpw, nks, waves, out=out)
            e1.record()
            torch.cuda.synchronize()
            if rnd:
                tt.append(e0.elapsed_time(e1) * 1000 / 12)
        out0 = out.clone()
        _lib.load().q3asr_set_option(2, 1)
        ops.skinny4(x16, M, Ws[0], N, K, 1, tpw, nks, waves, out=out)
        torch.cuda.synchronize()
        same0 = bool(torch.equal(out, out0)) if False else None
        same = ""
        if ref is None:
            ref = out.clone()
        else:
            same = f"  same bits as {geos[0]}: {bool(torch.equal(ref, out))}"
        t = sorted(ts)[len(ts) // 2]
        ops.skinny4(x16, M, Ws[2], N, K, 1, tpw, nks, waves, out=out)
        torch.cuda.synchronize()
        print(f"M={M:2d} geometry {g}: {t:6.1f} us  {N * K * 2 / t / 1e6:5.2f} TB/s   one workgroup per group {sorted(tt)[len(tt) // 2]:6.1f} us (same bits: {bool(torch.equal(out, out0))}){same}", flush=True)
