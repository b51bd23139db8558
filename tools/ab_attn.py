#!/usr/bin/env python3
"""A/B of attn_fwd launch forms in one process (q3asr_set_option): the prefill attention of a 32 x 408-token batch at the 1.7B shapes
(causal, 16 / 8 heads x 128, K / V read from the cache the RoPE kernel filled) and the encoder attention (32 clips x 4 windows, 16 heads x
64, bidirectional), HIP events around 20 back-to-back launches per arm, interleaved rounds; outputs compared bit for bit."""
import sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from qwen3_asr_mlx_amd import _lib, ops  # noqa: E402
dev = torch.device("cuda:0")
lib = _lib.load()
OPT = int(sys.argv[1]) if len(sys.argv) > 1 else 3
i32 = lambda v: torch.tensor(v, dtype=torch.int32, device=dev)      # noqa: E731


def bench(name, call, out):
    ts, outs = {0: [], 1: []}, {}
    for rnd in range(6):
        for arm in (1, 0):
            lib.q3asr_set_option(OPT, arm)
            state["i"] = 0
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                call()
            e1.record()
            torch.cuda.synchronize()
            if rnd:
                ts[arm].append(e0.elapsed_time(e1) * 1000 / 20)
            outs[arm] = out.clone()
    lib.q3asr_set_option(OPT, 1)
    med = lambda v: sorted(v)[len(v) // 2]      # noqa: E731
    print(f"{name:44s} option {OPT} off {med(ts[0]):7.1f} us   on {med(ts[1]):7.1f} us   ({100 * (med(ts[1]) / med(ts[0]) - 1):+.1f} %)  same bits: {bool(torch.equal(outs[0], outs[1]))}", flush=True)


# ---- prefill: B sequences of T tokens, caches [B][nkv][Lmax][hd] of several layers rotated (as the product walks 28 layers)
B, T, nh, nkv, hd, Lmax, NL = 32, 408, 16, 8, 128, 576, 6
q = [torch.randn(B * T, nh * hd, device=dev).to(torch.bfloat16) for _ in range(NL)]
kc = [torch.randn(B, nkv, Lmax, hd, device=dev).to(torch.bfloat16) for _ in range(NL)]
vc = [torch.randn(B, nkv, Lmax, hd, device=dev).to(torch.bfloat16) for _ in range(NL)]
out = torch.empty(B * T, nh * hd, dtype=torch.bfloat16, device=dev)
kv_off = torch.arange(B, dtype=torch.int64, device=dev) * (nkv * Lmax * hd)
starts, lens = i32([T * b for b in range(B)]), i32([T] * B)
state = {"i": 0}


def prefill():
    i = state["i"] = (state["i"] + 1) % NL
    ops.attn_fwd(ops.ptr(q[i]), nh * hd, ops.ptr(kc[i]), ops.ptr(vc[i]), kv_off, Lmax * hd, hd, out, nh * hd, starts, lens, B, T, nh, nkv, hd, True, hd ** -0.5)


bench(f"prefill attention {B} x {T}, {nh}/{nkv} x {hd}, causal", prefill, out)
# ---- encoder: 32 clips x 390 tokens in windows of 104 (4 per clip), 16 heads x 64, q | k | v interleaved per row
N, d, nhe = 32 * 390, 1024, 16
qkv = [torch.randn(N, 3 * d, device=dev).to(torch.bfloat16) for _ in range(NL)]
oute = torch.empty(N, d, dtype=torch.bfloat16, device=dev)
ws, wl = [], []
for b in range(32):
    for s0 in range(0, 390, 104):
        ws.append(b * 390 + s0)
        wl.append(min(104, 390 - s0))
kvo = torch.tensor(ws, dtype=torch.int64, device=dev) * (3 * d)
wsd, wld = i32(ws), i32(wl)


def encoder():
    i = state["i"] = (state["i"] + 1) % NL
    ops.attn_fwd(ops.ptr(qkv[i]), 3 * d, ops.off_ptr(qkv[i], d), ops.off_ptr(qkv[i], 2 * d), kvo, 64, 3 * d, oute, d, wsd, wld, len(ws), 104, nhe, nhe, 64, False, 64 ** -0.5)


bench(f"encoder attention {len(ws)} windows <= 104, {nhe} x 64", encoder, oute)
