#!/usr/bin/env python3
"""Where a workgroup of the prefill attention spends its time: wall_clock64() stamps (q3asr_attn_set_stamp) of single workgroups of
the 32 x 408-token causal launch at the 1.7B shapes -- heavy (7 key tiles), middle and light query tiles; medians over 9 launches."""
import sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from qwen3_asr_mlx_amd import _lib, ops  # noqa: E402
dev = torch.device("cuda:0")
lib = _lib.load()
i32 = lambda v: torch.tensor(v, dtype=torch.int32, device=dev)      # noqa: E731
B, T, nh, nkv, hd, Lmax = 32, 408, 16, 8, 128, 576
q = torch.randn(B * T, nh * hd, device=dev).to(torch.bfloat16)
kc = torch.randn(B, nkv, Lmax, hd, device=dev).to(torch.bfloat16)
vc = torch.randn(B, nkv, Lmax, hd, device=dev).to(torch.bfloat16)
out = torch.empty(B * T, nh * hd, dtype=torch.bfloat16, device=dev)
kv_off = torch.arange(B, dtype=torch.int64, device=dev) * (nkv * Lmax * hd)
starts, lens = i32([T * b for b in range(B)]), i32([T] * B)
buf = torch.zeros(64, dtype=torch.int64, device=dev)


def launch():
    ops.attn_fwd(ops.ptr(q), nh * hd, ops.ptr(kc), ops.ptr(vc), kv_off, Lmax * hd, hd, out, nh * hd, starts, lens, B, T, nh, nkv, hd, True, hd ** -0.5)


for _ in range(3):
    launch()
for wg in (0, 8, 1600, 1601, 1607, 1611, 3000, 3583):
    rows, qt, tot = [], None, []
    for rep in range(9):
        buf.zero_()
        lib.q3asr_attn_set_stamp(ops.ptr(buf), wg)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        launch()
        e1.record()
        torch.cuda.synchronize()
        lib.q3asr_attn_set_stamp(None, 0)
        v = buf.tolist()
        n, qt = v[0] & 0xffffffff, v[0] >> 32
        t = v[1: 1 + n]
        rows.append([(b - a) / 100.0 for a, b in zip(t, t[1:])])
        tot.append(e0.elapsed_time(e1) * 1e3)
    n = min(len(r) for r in rows)
    med = [sorted(r[i] for r in rows)[len(rows) // 2] for i in range(n)]
    tiles = (n - 1) // 4
    parts = [f"first tile staged {med[0]:.2f}"]
    for kt in range(tiles):
        seg = med[1 + 4 * kt: 1 + 4 * kt + 4] if kt else [None] + med[1:4]
        if kt == 0:
            parts.append(f"t0: S {med[1]:.2f} softmax {med[2]:.2f} PV {med[3]:.2f}")
        else:
            b = 4 * kt
            parts.append(f"t{kt}: stage {med[b]:.2f} S {med[b + 1]:.2f} softmax {med[b + 2]:.2f} PV {med[b + 3]:.2f}")
    parts.append(f"tail {med[-1]:.2f}")
    print(f"wg {wg:4d} (query tile {qt}, {tiles} key tiles): " + " | ".join(parts) + f"  = {sum(med):.2f} us in the workgroup; launch {sorted(tot)[len(tot) // 2]:.1f} us", flush=True)
