#!/usr/bin/env python3
"""In-situ tuning of the decode-step GEMM launch geometry: for each projection try every instantiated skinny4 geometry
(tiles per workgroup, k-steps per wave, waves) that fits its shape inside the real captured decode step (batch B, full-size
random weights) and time graph replays.  Coordinate descent, one GEMM at a time.  (Round 1's version tuned skinny3's
(n_splits, waves, tiles_per_wave, wg_ksplit); see git history.)  Usage: tune_decode.py [B] [tokens] [names|all] [bf16|fp8]"""
import sys
import time
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import qwen3_asr_mlx_amd as q  # noqa: E402
from qwen3_asr_mlx_amd import ops, synth  # noqa: E402
from qwen3_asr_mlx_amd.config import ModelConfig  # noqa: E402
from qwen3_asr_mlx_amd.generate import _merge, _side_stream  # noqa: E402
from qwen3_asr_mlx_amd.tokenizer import build_prompt  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
NT = int(sys.argv[2]) if len(sys.argv) > 2 else 96
dev = torch.device("cuda:0")
cfg = ModelConfig()
state = synth.random_state(cfg, seed=0, device=dev)
WDT = sys.argv[4] if len(sys.argv) > 4 else "bf16"
model = q.Qwen3ASR.from_state_dict(cfg, state, synth.SyntheticTokenizer(), dev, weight_dtype=WDT)
del state
dec = model._decoder
t = cfg.text_decoder
H, I = t.hidden_size, t.intermediate_size
qd, qkvd = t.num_attention_heads * t.head_dim, (t.num_attention_heads + 2 * t.num_key_value_heads) * t.head_dim
tok_counts = [390] * B
prompts = [build_prompt(n, [6364]) for n in tok_counts]
seq_lens = [len(p) for p in prompts]
feats = torch.randn(sum(tok_counts), t.hidden_size, device=dev)
sampling = {"temperature": 0.0, "top_p": 1.0, "top_k": 0, "repetition_penalty": 1.2, "repetition_context_size": 100, "seed": 0}


def measure():
    dec._ws.clear()
    Lmax = (max(seq_lens) + 128 + 63) // 64 * 64
    cache = dec.batch_cache(B, Lmax)
    ws = dec.workspace(B, 128)
    ws.n_gen.zero_(); ws.finished.zero_()
    ws.pos.copy_(torch.tensor([s - 1 for s in seq_lens], dtype=torch.int32))
    ws.max_new.copy_(torch.tensor([128] * B, dtype=torch.int32))
    emb = _merge(feats, prompts, tok_counts, dec)
    dec.prefill(emb, seq_lens, cache, logits_out=ws.logits)
    dec.sample_step(ws, ws.logits, sampling)
    side = _side_stream(dev)
    side.wait_stream(torch.cuda.current_stream())
    best = 1e9
    with torch.cuda.stream(side):
        dec.decode_step(ws, cache, sampling)
        dec.capture_step(ws, cache, sampling)
        for rep in range(2):
            side.synchronize()
            t0 = time.perf_counter()
            for _ in range(NT // 2):
                dec.replay_step(ws)
            side.synchronize()
            best = min(best, (time.perf_counter() - t0) / (NT // 2))
    torch.cuda.current_stream().wait_stream(side)
    return best * 1e6


def candidates(N, K, epi):
    """The instantiated skinny4 geometries (tiles per workgroup, k-steps per wave, waves) that fit the shape."""
    out = []
    for g in ops._SK4_BY_EPI[epi]:
        tpw, nks, waves = g
        if K % (32 * nks * waves) or (N // 16) % tpw or (WDT != "bf16" and nks % 2):
            continue
        if epi != 0 and K != 32 * nks * waves:
            continue
        if B > 32 and waves * tpw * 4 > 160:
            continue
        out.append(g)
    return out


table = ops._TUNED4 if WDT == "bf16" else ops._TUNED4_F8
base = measure()
print(f"B={B} {WDT}: current skinny4 table {base:.1f} us/token", flush=True)
shapes = [("qkv", qkvd, H, 0), ("o", H, qd, 0), ("gu", 2 * I, H, 2), ("down", H, I, 0), ("head", t.vocab_size, H, 1)]
if len(sys.argv) > 3 and sys.argv[3] != "all":
    shapes = [sh for sh in shapes if sh[0] in sys.argv[3].split(",")]
for name, N, K, epi in shapes:
    cur = table.get((N, K))
    res = []
    for c in candidates(N, K, epi):
        table[(N, K)] = c
        try:
            res.append((measure(), c))
        except Exception as e:                              # noqa: BLE001
            print("  ", c, "failed:", str(e)[:80])
    res.sort()
    print(f"{name} (N={N}, K={K}) current {cur}: " + ", ".join(f"{c} {u:.1f}" for u, c in res), flush=True)
    table[(N, K)] = res[0][1] if res and res[0][0] < base - 2 else cur
    base = min(base, res[0][0]) if res else base
print("final table:", dict(table))
print(f"final {measure():.1f} us/token")
