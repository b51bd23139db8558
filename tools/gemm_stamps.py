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


# ---- the 256x128 half-tile kernel (two workgroups per CU): shares of a K tile's loop segments, wave 0 of the middle workgroup
print("gemm2w_kernel loop segments (shader cycles per K tile, diagnostic build):")
buf2 = torch.zeros(16 + 4 * 8192, dtype=torch.int64, device=dev)


def census(buf, n_tiles, label):
    """Which workgroups shared a CU, and how much of a CU's busy time had two of them resident."""
    c = buf[16: 16 + 4 * n_tiles].reshape(n_tiles, 4).tolist()
    t_min = min(r[2] for r in c)
    cus = {}
    for b, (hw, xcc, t0, t1) in enumerate(c):
        key = (xcc & 0xF, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 0xF)      # xcc, se, sh, cu
        cus.setdefault(key, []).append((t0 - t_min, t1 - t_min, b))
    both = busy = 0
    first_pairs = []
    for key, iv in cus.items():
        ev = sorted([(a, 1) for a, _, _ in iv] + [(e, -1) for _, e, _ in iv])
        n = 0
        last = ev[0][0]
        for tt, d in ev:
            if n >= 1:
                busy += tt - last
            if n >= 2:
                both += tt - last
            n += d
            last = tt
        iv.sort()
        first_pairs.append((iv[0][2], iv[1][2] if len(iv) > 1 else -1, iv[1][0] - iv[0][0] if len(iv) > 1 else -1))
    dur = [(e - a) / 100.0 for iv in cus.values() for a, e, _ in iv]
    print(f"    census {label}: {len(cus)} CUs seen, {n_tiles} workgroups, two resident during {100.0 * both / max(1, busy):.0f} % of the CUs' busy time; "
          f"workgroup lifetime median {sorted(dur)[len(dur) // 2]:.1f} us; first two blocks on a CU (block, block, start offset in 10 ns): {first_pairs[:6]}")

for (N, K, name) in [(4096, 1024, "fc1 shape N=4096 K=1024"), (2048, 6144, "down shape N=2048 K=6144"), (12288, 2048, "gate/up shape N=12288 K=2048")]:
    A = torch.randn(M, K, device=dev).to(torch.bfloat16)
    W = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
    out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    _lib.call("q3asr_gemm_force_kernel", 2)
    try:
        for _ in range(3):
            ops.gemm(A, W, M, N, K, out)
        rows = []
        for rep in range(5):
            buf2.zero_()
            _lib.call("q3asr_gemm_set_stamp", ops.ptr(buf2))
            ops.gemm(A, W, M, N, K, out)
            torch.cuda.synchronize()
            _lib.call("q3asr_gemm_set_stamp", None)
            rows.append(buf2.tolist()[:9])
        census(buf2, ((M + 255) // 256) * ((N + 127) // 128), name)
    finally:
        _lib.call("q3asr_gemm_force_kernel", 0)
    med = [sorted(r[i] for r in rows)[len(rows) // 2] for i in range(9)]
    nk = K // 64
    names = ["frag reads", "MFMA block 1", "mid barrier", "DMA issue", "MFMA block 2", "vmcnt + barrier"]
    print(f"  {name}: " + ", ".join(f"{n} {med[i] / nk:.0f}" for i, n in enumerate(names)) +
          f" | per K tile {sum(med[:6]) / nk:.0f} (64 MFMAs = 1024 at the issue rate); prologue {med[6]}, epilogue {med[7]}")
