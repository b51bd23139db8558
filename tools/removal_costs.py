#!/usr/bin/env python3
"""What each kind of decode-step kernel really costs inside the captured chain: re-capture the step with all launches of
one kind skipped (results are garbage; timing only) and compare graph-replay time per token.  rocprofv3 over-charges
small kernels, this does not.  Usage: removal_costs.py [B]"""
import sys
import time
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import qwen3_asr_mlx_amd as q  # noqa: E402
from qwen3_asr_mlx_amd import ops, synth, _lib  # noqa: E402
from qwen3_asr_mlx_amd.config import ModelConfig  # noqa: E402
from qwen3_asr_mlx_amd.generate import _merge, _side_stream  # noqa: E402
from qwen3_asr_mlx_amd.tokenizer import build_prompt  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0")
cfg = ModelConfig()
state = synth.random_state(cfg, seed=0, device=dev)
model = q.Qwen3ASR.from_state_dict(cfg, state, synth.SyntheticTokenizer(), dev)
del state
dec = model._decoder
t = cfg.text_decoder
tok_counts = [390] * B
prompts = [build_prompt(n, [6364]) for n in tok_counts]
seq_lens = [len(p) for p in prompts]
feats = torch.randn(sum(tok_counts), t.hidden_size, device=dev)
sampling = {"temperature": 0.0, "top_p": 1.0, "top_k": 0, "repetition_penalty": 1.2, "repetition_context_size": 100, "seed": 0}
skip: set = set()
real_call = ops.call


def call(name, *a):
    if name in skip or (name in ("q3asr_skinny3_gemm", "q3asr_skinny4_gemm") and ("gemm:%d" % a[4]) in skip):      # a[4] = N
        return
    if name in ("q3asr_skinny3_gemm_norm", "q3asr_skinny4_gemm_norm") and ("gemm:%d" % a[10]) in skip:                  # fused small-batch form: a[10] = N
        return
    return real_call(name, *a)


ops.call = call


def measure(nt=96):
    dec._ws.clear()
    Lmax = (max(seq_lens) + 128 + 63) // 64 * 64
    cache = dec.batch_cache(B, Lmax)
    ws = dec.workspace(B, 128)
    ws.n_gen.zero_(); ws.finished.zero_()
    ws.pos.copy_(torch.tensor([s - 1 for s in seq_lens], dtype=torch.int32))
    ws.max_new.copy_(torch.tensor([128] * B, dtype=torch.int32))
    saved = set(skip)
    skip.clear()
    emb = _merge(feats, prompts, tok_counts, dec)
    dec.prefill(emb, seq_lens, cache, logits_out=ws.logits)
    dec.sample_step(ws, ws.logits, sampling)
    side = _side_stream(dev)
    side.wait_stream(torch.cuda.current_stream())
    best = 1e9
    with torch.cuda.stream(side):
        dec.decode_step(ws, cache, sampling)
        skip.update(saved)
        dec.capture_step(ws, cache, sampling)
        for _ in range(2):
            side.synchronize()
            t0 = time.perf_counter()
            for _ in range(nt // 2):
                dec.replay_step(ws)
            side.synchronize()
            best = min(best, (time.perf_counter() - t0) / (nt // 2))
    torch.cuda.current_stream().wait_stream(side)
    return best * 1e6


H, I = t.hidden_size, t.intermediate_size
qkvd = (t.num_attention_heads + 2 * t.num_key_value_heads) * t.head_dim
base = measure()
print(f"B={B}: full step {base:.1f} us/token (context {seq_lens[0]}..{seq_lens[0] + 96})")
kinds = [("attention (28)", {"q3asr_attn_decode_fused"}), ("residual+norm (56)", {"q3asr_slab_resnorm_x16"}),
         ("qkv GEMM (28)", {"gemm:%d" % qkvd}), ("o + down GEMMs (56)", {"gemm:%d" % H}), ("gate/up GEMM (28)", {"gemm:%d" % (2 * I)}),
         ("tied head (1)", {"gemm:%d" % t.vocab_size}), ("sampling (1)", {"q3asr_sample_step"}), ("embed+norm (1)", {"q3asr_embed_norm_x16"})]
tot = 0.0
for name, names in kinds:
    skip.clear(); skip.update(names)
    u = measure()
    skip.clear()
    n = int(name.split("(")[1].rstrip(")"))
    print(f"  without {name:22s}: {u:8.1f} us/token  -> {base - u:7.1f} us per step, {(base - u) / n:6.2f} us per launch")
    tot += base - u
print(f"  sum of removals {tot:.1f} us of {base:.1f}")
