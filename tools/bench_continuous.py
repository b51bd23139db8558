#!/usr/bin/env python3
"""Ragged request at the 1.7B shapes: N clips of 5-30 s with token budgets proportional to their length (4.27 tokens per
audio-second, 128 for a 30 s clip), decoded (a) in fixed batches of MAX_BATCH, two in flight, and (b) through MAX_BATCH rows
that are refilled as sequences end (generate_continuous), on one and on two lanes.  Prints audio-s / wall-s for each.
Usage: bench_continuous.py [n_clips] [max_batch]"""
import json
import sys
import time
from pathlib import Path
import numpy as np
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import qwen3_asr_mlx_amd as q  # noqa: E402
from qwen3_asr_mlx_amd import synth  # noqa: E402
from qwen3_asr_mlx_amd.config import ModelConfig  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 192
max_batch = int(sys.argv[2]) if len(sys.argv) > 2 else 64
workload = sys.argv[3] if len(sys.argv) > 3 else "proportional"     # or "unpredictable": 30 s clips, budgets unrelated to length
dev = torch.device("cuda:0")
cfg = ModelConfig()
state = synth.random_state(cfg, seed=0, device=dev)
model = q.Qwen3ASR.from_state_dict(cfg, state, synth.SyntheticTokenizer(), dev)
del state
model.MAX_BATCH = max_batch
rng = np.random.default_rng(7)
durs = rng.uniform(5.0, 30.0, n) if workload == "proportional" else rng.uniform(28.0, 30.0, n)
buds = rng.integers(16, 129, n)
jobs = []
for i, d in enumerate(durs):
    x = (0.1 * np.random.default_rng(1234 + i).standard_normal(int(d * 16000))).astype(np.float32)
    jobs.append((0, x, max(1, int(round(d * 128 / 30))) if workload == "proportional" else int(buds[i])))
audio_s = float(durs.sum())
args = ("English", 0.0, 1.0, 0, 1.2, 100)


def timed(label, reps=3):
    model._run_jobs(jobs, *args)                  # warm-up: plans, graphs
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        texts = model._run_jobs(jobs, *args)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    best = min(ts)
    return {"mode": label, "wall_s": round(best, 4), "rtf": round(audio_s / best, 1)}, texts


out = {"workload": workload, "clips": n, "audio_s": round(audio_s, 1), "tokens": sum(j[2] for j in jobs), "rows": max_batch}
model.continuous, model.lanes = False, 1
r0, t0 = timed("fixed batches, one at a time")
model.lanes = 2
r1, t1 = timed("fixed batches, two in flight")
model.continuous, model.lanes = True, 1
sweep = []
for ma in (1, 4, 8, 16, 32):
    model.min_admit = ma
    r, t = timed(f"refilled rows, one lane, min_admit {ma}", reps=2)
    r["texts_equal_fixed"] = t == t0
    r["steps"] = model.last_continuous_stats[0]["steps"]
    sweep.append(r)
best = max(sweep, key=lambda r: r["rtf"])
model.min_admit = int(best["mode"].rsplit(" ", 1)[1])
r2, t2 = timed("refilled rows, one lane, best min_admit")
s2 = [dict(s) for s in model.last_continuous_stats]
model.continuous_lanes = 2
r3, t3 = timed("refilled rows, two lanes, best min_admit")
s3 = [dict(s) for s in model.last_continuous_stats]
# the queue front door: every job submitted at once from a client thread, results awaited
from qwen3_asr_mlx_amd.serve import ContinuousTranscriber  # noqa: E402
fd = []
for enc_thread in (False, True):
    srv = ContinuousTranscriber(model, rows=max_batch, chunk_duration=60.0, max_new_tokens=256, min_admit=model.min_admit, encode_thread=enc_thread)
    for f in [srv.submit(j[1], max_tokens=j[2]) for j in jobs[:max_batch]]:      # warm-up: plans, graph
        f.result()
    ts = []
    for _ in range(3):
        torch.cuda.synchronize()
        q0 = time.perf_counter()
        futs = [srv.submit(j[1], max_tokens=j[2]) for j in jobs]
        t4 = [f.result().text for f in futs]
        ts.append(time.perf_counter() - q0)
    srv.close()
    fd.append({"mode": "queue front door (ContinuousTranscriber), all jobs submitted at once, encoder " + ("on its own thread and stream" if enc_thread else "between decode chunks"),
               "wall_s": round(min(ts), 4), "rtf": round(audio_s / min(ts), 1), "texts_equal_fixed": t4 == t0})
out["runs"] = [r0, r1] + sweep + [r2, r3] + fd
out["texts_equal"] = t0 == t1 == t2 == t3
out["differing"] = {"fixed2_vs_fixed1": sum(a != b for a, b in zip(t0, t1)), "refill1_vs_fixed1": sum(a != b for a, b in zip(t0, t2)),
                    "refill2_vs_fixed1": sum(a != b for a, b in zip(t0, t3)), "first": [i for i, (a, b) in enumerate(zip(t0, t2)) if a != b][:5]}
out["refill_stats_one_lane"], out["refill_stats_two_lanes"] = s2, s3
print(json.dumps(out))
