#!/usr/bin/env python3
"""Batch-1 request: how much of mel + encoder + prefill is the host issuing launches?  Wall time (host clock, synchronised) beside the
device time between the first and the last kernel of the stage (HIP events on the launch stream) and the number of C-ABI calls.
Usage: python3 tools/b1_front.py [batch=1] [seconds=30]"""
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
from qwen3_asr_mlx_amd.generate import _merge  # noqa: E402
from qwen3_asr_mlx_amd.tokenizer import build_prompt  # noqa: E402
from signals import signal  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 30.0
dev = torch.device("cuda:0")
cfg = ModelConfig()
model = q.Qwen3ASR.from_state_dict(cfg, synth.random_state(cfg, seed=0, device=dev), synth.SyntheticTokenizer(), dev)
dec, enc = model._decoder, model._encoder
clips = [signal("speechlike", int(secs * 16000), seed=1234 + i) for i in range(B)]
fe = q.audio.get_frontend(dev)
calls = [0]
orig_call = ops.call


def counting_call(*a, **k):
    calls[0] += 1
    return orig_call(*a, **k)


def stage(name, fn, reps=7):
    out = None
    rows = []
    for r in range(reps):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        calls[0] = 0
        t0 = time.perf_counter()
        e0.record()
        out = fn()
        e1.record()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        rows.append((1e3 * (t2 - t0), 1e3 * (t1 - t0), e0.elapsed_time(e1), calls[0]))
    rows.sort()
    w, h, d, n = rows[len(rows) // 2]
    print(f"{name:10s} wall {w:6.2f} ms | host issue {h:6.2f} ms | device span {d:6.2f} ms | {n} C-ABI calls", flush=True)
    return out


ops.call = counting_call
for mod in (q.audio, q.encoder, q.decoder):
    if getattr(mod, "ops", None) is ops:
        pass
audio = stage("pack+H2D", lambda: q.audio.pack_audio(clips, dev))
mel = stage("mel", lambda: fe(*audio))
feats = stage("encoder", lambda: enc.forward_batch(*mel))
prompts = [build_prompt(c, [6364]) for c in feats[1]]
seq = [len(p) for p in prompts]
emb = stage("merge", lambda: _merge(feats[0], prompts, feats[1], dec))
cache = dec.batch_cache(B, (max(seq) + 128 + 63) // 64 * 64)
ws = dec.workspace(B, 128)
stage("prefill", lambda: dec.prefill(emb, seq, cache, logits_out=ws.logits))
t0 = time.perf_counter()
for _ in range(5):
    model.transcribe_batch(clips, max_tokens=128)
torch.cuda.synchronize()
print(f"whole request (transcribe_batch, 128 tokens): {1e3 * (time.perf_counter() - t0) / 5:.1f} ms")
