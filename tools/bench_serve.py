#!/usr/bin/env python3
"""Open-loop load on the queue front door (serve.ContinuousTranscriber) at the 1.7B shapes: 30 s clips, 128 tokens each,
Poisson arrivals at several rates; per-request latency (submit -> result) and sustained throughput.
Usage: bench_serve.py [rows] [rate,rate,...] [seconds] [async|sync] [min_admit|-] [check_every]"""
import json
import sys
import threading
import time
from pathlib import Path
import numpy as np
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import qwen3_asr_mlx_amd as q  # noqa: E402
from qwen3_asr_mlx_amd import synth  # noqa: E402
from qwen3_asr_mlx_amd.config import ModelConfig  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 32
rates = [float(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [10, 25, 50, 80, 110]
seconds = float(sys.argv[3]) if len(sys.argv) > 3 else 6.0
async_admit = len(sys.argv) > 4 and sys.argv[4] == "async"
min_admit = int(sys.argv[5]) if len(sys.argv) > 5 and sys.argv[5] != "-" else None
check_every = int(sys.argv[6]) if len(sys.argv) > 6 else None
dev = torch.device("cuda:0")
cfg = ModelConfig()
state = synth.random_state(cfg, seed=0, device=dev)
model = q.Qwen3ASR.from_state_dict(cfg, state, synth.SyntheticTokenizer(), dev)
del state
clips = [(0.1 * np.random.default_rng(1234 + i).standard_normal(480000)).astype(np.float32) for i in range(16)]
NTOK = 128
t0 = time.perf_counter()
one = [model.transcribe(clips[0], max_tokens=NTOK) for _ in range(4)]
t_one = (time.perf_counter() - t0) / 4
out = {"async_admit": async_admit, "min_admit": min_admit, "check_every": check_every, "rows": rows, "clip_seconds": 30, "tokens": NTOK, "transcribe_one_at_a_time_ms": round(1e3 * t_one, 1), "runs": []}
kw = {} if check_every is None else {"check_every": check_every}
srv = q.ContinuousTranscriber(model, rows=rows, chunk_duration=60.0, max_new_tokens=NTOK, async_admit=async_admit, min_admit=min_admit, **kw)
for f in [srv.submit(clips[i % 16], max_tokens=NTOK) for i in range(rows)]:        # warm-up: plans, graph
    f.result()
for rate in rates:
    rng = np.random.default_rng(int(rate * 10))
    lat, lock = [], threading.Lock()
    futs = []
    start = time.perf_counter()
    t_next = start
    n = 0
    while True:
        t_next += rng.exponential(1.0 / rate)
        if t_next - start > seconds:
            break
        dt = t_next - time.perf_counter()
        if dt > 0:
            time.sleep(dt)
        ts = time.perf_counter()
        f = srv.submit(clips[n % 16], max_tokens=NTOK)

        def done(_f, ts=ts):
            with lock:
                lat.append(time.perf_counter() - ts)
        f.add_done_callback(done)
        futs.append(f)
        n += 1
    for f in futs:
        f.result()
    wall = time.perf_counter() - start
    a = np.sort(np.array(lat))
    out["runs"].append({"offered_req_per_s": rate, "requests": n, "completed_req_per_s": round(n / wall, 1), "rtf": round(30.0 * n / wall, 1),
                        "latency_ms": {"p50": round(1e3 * float(a[len(a) // 2]), 1), "p95": round(1e3 * float(a[int(len(a) * 0.95)]), 1),
                                       "max": round(1e3 * float(a[-1]), 1)}})
    print(json.dumps(out["runs"][-1]), flush=True)
srv.close()
out["scheduler"] = srv.stats
print(json.dumps(out))
