#!/usr/bin/env python3
"""BASELINE.json configs[4] (fp8 weights, 20 min long-form, chunk_duration 30, T = 0.7 / top-p 0.9): how the 40 segments are cut
into decode batches, one process, interleaved.  Arms: one batch of 40 (shipped), two batches of 20 in flight on two lanes, 32 + 8 in
flight; and the greedy sampler in place of the draw (what the sampler costs).  Also host-side stamps of one run."""
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
wd = sys.argv[1] if len(sys.argv) > 1 else "fp8"
ndec = int(sys.argv[2]) if len(sys.argv) > 2 else 128
m8 = q.Qwen3ASR.from_state_dict(cfg, state, synth.SyntheticTokenizer(), dev, weight_dtype=wd)
del state
audio = np.concatenate([signal("speechlike", 16000 * 60, seed=100 + i) for i in range(20)])


def run(temperature=0.7, top_p=0.9, stamps=None):
    t0 = time.perf_counter()
    pts = _find_split_points(audio, 30 * 16000, 5 * 16000)
    t1 = time.perf_counter()
    jobs, prev = [], 0
    for sp in pts + [len(audio)]:
        seg = audio[prev:sp]
        prev = sp
        if len(seg):
            jobs.append((0, seg, ndec))
    texts = m8._run_jobs(jobs, "English", temperature, top_p, 0, 1.2, 100)
    torch.cuda.synchronize()
    if stamps is not None:
        stamps.update(split_ms=1e3 * (t1 - t0), rest_ms=1e3 * (time.perf_counter() - t1))
    return texts


arms = {"one batch of 40": dict(MAX_BATCH=64, lanes=2), "2 x 20 in flight": dict(MAX_BATCH=20, lanes=2), "32 + 8 in flight": dict(MAX_BATCH=32, lanes=2),
        "4 x 10 in flight": dict(MAX_BATCH=10, lanes=4), "one batch of 40, greedy sampler": dict(MAX_BATCH=64, lanes=2, greedy=True)}
res = {k: [] for k in arms}
for rnd in range(4):
    for name, a in arms.items():
        m8.MAX_BATCH, m8.lanes = a["MAX_BATCH"], a["lanes"]
        torch.cuda.synchronize()
        s0 = time.perf_counter()
        if a.get("greedy"):
            run(0.0, 1.0)
        else:
            run()
        dt = time.perf_counter() - s0
        if rnd:
            res[name].append(dt)
for name, v in res.items():
    med = sorted(v)[len(v) // 2]
    print(f"{wd} ndec={ndec} {name:34s}: {1e3 * med:8.1f} ms  RTF {1200.0 / med:6.0f}x   (runs: {' '.join(f'{1e3 * x:.0f}' for x in v)})", flush=True)
m8.MAX_BATCH, m8.lanes = 64, 2
st = {}
run(stamps=st)
print("host stamps of one run:", {k: round(v, 1) for k, v in st.items()})
