#!/usr/bin/env python3
"""Do results depend on what freed device memory holds?  Runs the same ragged batch with the allocator's free blocks
filled with NaN bit patterns, then with zeros, and compares stage by stage.  Usage: poison_check.py [B] [small|full]"""
import sys
from pathlib import Path
import numpy as np
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import qwen3_asr_mlx_amd as q  # noqa: E402
from qwen3_asr_mlx_amd import synth  # noqa: E402
from qwen3_asr_mlx_amd.audio import get_frontend, pack_audio  # noqa: E402
from qwen3_asr_mlx_amd.config import ModelConfig  # noqa: E402
from qwen3_asr_mlx_amd.generate import generate_batch  # noqa: E402
from qwen3_asr_mlx_amd.tokenizer import build_prompt  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
full = (sys.argv[2] if len(sys.argv) > 2 else "full") == "full"
dev = torch.device("cuda:0")
cfg = ModelConfig() if full else synth.small_config()
state = synth.random_state(cfg, seed=0, device=dev)
model = q.Qwen3ASR.from_state_dict(cfg, state, synth.SyntheticTokenizer(), dev)
del state
rng = np.random.default_rng(7)
durs = np.sort(rng.uniform(5.0, 30.0, B))[::-1]
xs = [(0.1 * np.random.default_rng(1234 + i).standard_normal(int(d * 16000))).astype(np.float32) for i, d in enumerate(durs)]
budgets = [max(1, int(round(d * 128 / 30))) for d in durs]


def poison(byte):
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    blocks = [torch.full((1 << 30,), byte, dtype=torch.uint8, device=dev) for _ in range(12)]
    small = [torch.full((n,), byte, dtype=torch.uint8, device=dev) for n in (1 << 12, 1 << 16, 1 << 20, 1 << 24) for _ in range(16)]
    torch.cuda.synchronize()
    del blocks, small


def run():
    audio, lens_dev, lens = pack_audio(xs, dev)
    mel, n_frames = get_frontend(dev)(audio, lens_dev, lens)
    feats, counts = model._encoder.forward_batch(mel, n_frames)
    lang_ids = model._tokenizer.encode(" English")
    prompts = [build_prompt(n, lang_ids) for n in counts]
    toks, ws, cache = generate_batch(model._decoder, feats, counts, prompts, budgets, repetition_penalty=1.2, return_state=True)
    valid_mel = [mel[b, :, :n].clone() for b, n in enumerate(n_frames)]
    return valid_mel, feats.clone(), toks


out = {}
for name, byte in (("nan", 0xFF), ("zero", 0x00), ("nan2", 0x7F)):
    model._decoder._ws.clear(); model._decoder._caches.clear()
    poison(byte)
    out[name] = run()
    print(name, "done", flush=True)
a = out["nan"]
for name in ("zero", "nan2"):
    b = out[name]
    mel_eq = all(torch.equal(x, y) for x, y in zip(a[0], b[0]))
    feats_eq = torch.equal(a[1], b[1])
    bad_rows = (a[1] != b[1]).any(dim=1).nonzero().flatten().tolist() if not feats_eq else []
    ntok = sum(x != y for x, y in zip(a[2], b[2]))
    print(f"nan vs {name}: mel equal {mel_eq}; encoder features equal {feats_eq} (rows differing: {len(bad_rows)}, first {bad_rows[:8]}); "
          f"sequences with different tokens {ntok} of {B}; any NaN in features {bool(torch.isnan(a[1]).any())}/{bool(torch.isnan(b[1]).any())}", flush=True)
