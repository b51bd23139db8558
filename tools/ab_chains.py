#!/usr/bin/env python3
"""A/B in one process, interleaved rounds: the decode loop of a batch as ONE launch chain (the product) against 2 (and 4)
chains of B/2 (B/4) rows, each with its own captured graph on its own stream (the round-2 review's route (b): "the 32 rows
as two 16-row chains ... so that the second chain's weight read of a layer is served from the 256 MiB Infinity Cache").
The chains are views of the parent workspace's per-row state and of the K/V rows; the kernels are batch-invariant and the
sampler keys its draws by the request-wide row index, so the tokens must be -- and are -- identical.
Measured (MI355X, 32 x 30 s, 128 tokens, profiles/r03_ab_chains.txt): 249.8 ms as one chain, 316.7 ms as two, 559.8 ms as
four: every launch of a decode step already fills the chip, so two chains do not fill each other's gaps, they read the
weights twice (the second, non-temporal read is not served faster from the Infinity Cache).  Not in the product.
Usage: ab_chains.py [B=32] [seconds=30] [ndec=128] [rounds=6] [chains=1,2,4] [weights=bf16]"""
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import qwen3_asr_mlx_amd as q  # noqa: E402
from qwen3_asr_mlx_amd import synth  # noqa: E402
from qwen3_asr_mlx_amd.config import ModelConfig  # noqa: E402
from signals import signal  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 30.0
ndec = int(sys.argv[3]) if len(sys.argv) > 3 else 128
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 6
variants = [int(v) for v in (sys.argv[5] if len(sys.argv) > 5 else "1,2,4").split(",")]
weights = sys.argv[6] if len(sys.argv) > 6 else "bf16"
dev = torch.device("cuda:0")
cfg = ModelConfig()
model = q.Qwen3ASR.from_state_dict(cfg, synth.random_state(cfg, seed=0, device=dev), synth.SyntheticTokenizer(), dev, weight_dtype=weights)
clips = [signal("speechlike", int(secs * 16000), seed=1234 + i) for i in range(B)]
audio, lens_dev, lens = q.audio.pack_audio(clips, dev)
dec = model._decoder

from qwen3_asr_mlx_amd import _lib  # noqa: E402
import importlib  # noqa: E402
gen = importlib.import_module("qwen3_asr_mlx_amd.generate")
from qwen3_asr_mlx_amd.decoder import DecodeWorkspace, KVCache  # noqa: E402
N_CHAINS = 1


def chains_of(ws, cache, n):
    store = ws.__dict__.setdefault("_chains", {})
    key = (cache.k.data_ptr(), cache.max_len, n)
    if key not in store:
        out = []
        bounds = [(ws.B * c) // n for c in range(n + 1)]
        for c in range(n):
            r0, r1 = bounds[c], bounds[c + 1]
            cws = DecodeWorkspace(dec.config, r1 - r0, ws.max_out, dev, fp8=weights != "bf16", parent=ws, row0=r0)
            cc = KVCache()
            cc.k, cc.v = cache.k[:, r0:r1], cache.v[:, r0:r1]         # [layers][rows][n_kv][Lmax][hd]: view[li] is contiguous
            cc.batch, cc.max_len, cc.offset = r1 - r0, cache.max_len, cache.offset
            nbytes = _lib.load().q3asr_attn_decode_fused_workspace(r1 - r0, dec.config.num_key_value_heads, cache.max_len)
            cc.attn_ws = torch.zeros((nbytes // 4,), dtype=torch.float32, device=dev)
            out.append((cws, cc))
        store[key] = out
    return store[key]


def decode_chained(ws, cache, sampling, steps, n):
    parts = chains_of(ws, cache, n)
    streams = [gen._side_stream(dev, c) for c in range(n)]
    cur = torch.cuda.current_stream()
    for st in streams:
        st.wait_stream(cur)
    for (cws, cc), st in zip(parts, streams):
        with torch.cuda.stream(st):
            dec.decode_step(cws, cc, sampling)
            if steps > 1:
                dec.capture_step(cws, cc, sampling)
    done = [1] * n
    while min(done) < steps:
        for c, ((cws, cc), st) in enumerate(zip(parts, streams)):    # one launch per chain per turn: the chains stay in step
            if done[c] < steps:
                with torch.cuda.stream(st):
                    done[c] += dec.replay_step(cws, multi=steps - done[c] >= dec.MULTI_STEPS)
    for st in streams:
        cur.wait_stream(st)


def generate_chained(audio, lens_dev, lens, budgets):
    fe = q.audio.get_frontend(dev)
    mel, n_frames = fe(audio, lens_dev, lens)
    feats, counts = model._encoder.forward_batch(mel, n_frames)
    from qwen3_asr_mlx_amd.tokenizer import build_prompt
    prompts = [build_prompt(n_, [6364]) for n_ in counts]
    seq = [len(p) for p in prompts]
    embeds = gen._merge(feats, prompts, counts, dec)
    Lmax = (max(t + m for t, m in zip(seq, budgets)) + 63) // 64 * 64
    cache = dec.batch_cache(B, Lmax)
    ws = dec.workspace(B, max(budgets))
    ws.n_gen.zero_(); ws.finished.zero_()
    ws.pos.copy_(torch.tensor([t - 1 for t in seq], dtype=torch.int32))
    ws.max_new.copy_(torch.tensor(budgets, dtype=torch.int32))
    sampling = {"temperature": 0.0, "top_p": 1.0, "top_k": 0, "repetition_penalty": 1.2, "repetition_context_size": 100, "seed": 0}
    dec.prefill(embeds, seq, cache, logits_out=ws.logits)
    dec.sample_step(ws, ws.logits, sampling)
    decode_chained(ws, cache, sampling, max(budgets) - 1, N_CHAINS)
    n_gen = ws.n_gen.tolist()
    toks = ws.out_tokens.cpu()
    return [toks[b, : n_gen[b]].tolist() for b in range(B)]


def run(chains):
    global N_CHAINS
    N_CHAINS = chains
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if chains == 1:
        toks = model.generate_tokens_device(audio, lens_dev, lens, [ndec] * B, "English", repetition_penalty=1.2)
    else:
        toks = generate_chained(audio, lens_dev, lens, [ndec] * B)
    torch.cuda.synchronize()
    return toks, (time.perf_counter() - t0) * 1e3


ref = None
times = {v: [] for v in variants}
for v in variants:                                      # warm-up: code objects, graphs (the 4-step graph comes with the second use)
    for _ in range(3):
        toks, _ = run(v)
    ref = ref or toks
    print(f"chains={v}: tokens equal to chains={variants[0]}: {toks == ref}", flush=True)
for r in range(rounds):
    for v in variants:
        toks, ms = run(v)
        assert toks == ref, f"chains={v} changed the tokens"
        times[v].append(ms)
for v in variants:
    t = sorted(times[v])
    print(f"chains={v}: median {t[len(t) // 2]:.1f} ms, min {t[0]:.1f} ms per request of {B} x {secs:g} s, {ndec} tokens "
          f"({(t[len(t) // 2] - 0) :.1f})", flush=True)
