#!/usr/bin/env python3
"""BASELINE.json configs[4] as the bench's `longform_fp8` leg runs it (fp8 decoder weights, 20 min of audio as one request, 40 segments
decoded as one batch, T = 0.7 / top-p 0.9 / penalty 1.2, 128 tokens per segment), three times, for a profiler to wrap:
    rocprofv3 --kernel-trace --stats --output-format csv -d <dir> -- python3 tools/config5_probe.py"""
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
m8 = q.Qwen3ASR.from_state_dict(cfg, synth.random_state(cfg, seed=0, device=dev), synth.SyntheticTokenizer(), dev, weight_dtype="fp8")
audio = np.concatenate([signal("speechlike", 16000 * 60, seed=100 + i) for i in range(20)])
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pts = _find_split_points(audio, 30 * 16000, 5 * 16000)
    jobs, prev = [], 0
    for sp in pts + [len(audio)]:
        seg, prev = audio[prev:sp], sp
        if len(seg):
            jobs.append((0, seg, 128))
    m8._run_jobs(jobs, "English", 0.7, 0.9, 0, 1.2, 100)
    torch.cuda.synchronize()
    print(f"[config5_probe] request {rep}: {1e3 * (time.perf_counter() - t0):.1f} ms, {len(jobs)} segments", flush=True)
