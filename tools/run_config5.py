#!/usr/bin/env python3
"""BASELINE.json configs[4]: Qwen3-ASR-1.7B shapes, fp8 decoder weights, 20 min of synthetic audio split at
low-energy points into <= 30 s segments (chunk_duration=30), temperature 0.7 / top-p 0.9, one MI355X.
Segments are decoded as one batch; with random weights no EOS appears, so every segment decodes its full budget
max(256, 50 * seconds) = 1500 tokens (the reference's own budget rule, model.py:416)."""
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import qwen3_asr_mlx_amd as q  # noqa: E402
from qwen3_asr_mlx_amd import synth  # noqa: E402
from qwen3_asr_mlx_amd.config import ModelConfig  # noqa: E402
from qwen3_asr_mlx_amd.model import _find_split_points  # noqa: E402
from signals import signal  # noqa: E402

dev = torch.device("cuda:0")
cfg = ModelConfig()
state = synth.random_state(cfg, seed=0, device=dev)
model = q.Qwen3ASR.from_state_dict(cfg, state, synth.SyntheticTokenizer(), dev, weight_dtype=sys.argv[1] if len(sys.argv) > 1 else "fp8")
del state
minutes = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
audio = np.concatenate([signal("speechlike", 16000 * 60, seed=100 + i) for i in range(int(minutes))])
pts = _find_split_points(audio, 30 * 16000, 5 * 16000)
print(f"{len(audio) / 16000:.0f} s of audio, {len(pts) + 1} segments", flush=True)
model.transcribe(audio[: 16000 * 31], temperature=0.7, top_p=0.9, chunk_duration=30.0, max_tokens=8)   # warm
for rep in range(2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = model.transcribe(audio, temperature=0.7, top_p=0.9, chunk_duration=30.0)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"run {rep}: {dt:.2f} s wall, RTF {r.duration / dt:.0f}x, text length {len(r.text)}", flush=True)
