"""Where does `end_to_end - value` go?  Repeats the public transcribe_batch (host arrays in, texts out) on the bench's request with
Python's cyclic GC enabled / disabled / frozen, both staging forms, and prints every run (not a median): a collection of the oldest
generation landing inside a request shows up as one slow run in a fixed position of the cycle.
Usage: python tools/e2e_probe.py [runs=8]"""
import gc
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import qwen3_asr_mlx_amd as q  # noqa: E402
from qwen3_asr_mlx_amd import audio as qaudio, synth  # noqa: E402
from qwen3_asr_mlx_amd.config import ModelConfig  # noqa: E402
from signals import signal  # noqa: E402


def main():
    runs = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    dev = torch.device("cuda:0")
    cfg = ModelConfig()
    state = synth.random_state(cfg, seed=0, device=dev)
    model = q.Qwen3ASR.from_state_dict(cfg, state, synth.SyntheticTokenizer(), dev)
    del state
    clips = [signal("speechlike", 480000, seed=1234 + i) for i in range(32)]
    for _ in range(3):
        model.transcribe_batch(clips, max_tokens=128)
    torch.cuda.synchronize()

    def series(label, mode):
        qaudio.PACK_MODE = mode
        out = []
        for _ in range(runs):
            c0 = gc.get_count()
            st = qaudio.PACK_STATS = {}
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            model.transcribe_batch(clips, max_tokens=128)
            torch.cuda.synchronize()
            dt = 1e3 * (time.perf_counter() - t0)
            qaudio.PACK_STATS = None
            s = qaudio.pack_stats_finish(st)
            out.append(dt)
            print(f"  {label:28s} {mode:8s} {dt:7.1f} ms   fill {s['host_fill_ms']:5.2f}  span {s['copy_span_ms']:5.2f}  gc counts before {c0}", flush=True)
        print(f"{label:30s} {mode:8s} median {np.median(out):7.1f}  min {min(out):7.1f}  max {max(out):7.1f}", flush=True)

    gc_stats0 = [dict(s) for s in gc.get_stats()]
    for mode in ("grouped", "single"):
        series("gc enabled", mode)
    print("gc collections during the enabled series:", [b["collections"] - a["collections"] for a, b in zip(gc_stats0, gc.get_stats())], flush=True)
    gc.collect()
    gc.freeze()
    for mode in ("grouped", "single"):
        series("gc frozen (gc.freeze)", mode)
    gc.disable()
    for mode in ("grouped", "single"):
        series("gc disabled", mode)
    gc.enable()
    t0 = time.perf_counter()
    gc.unfreeze()
    n = gc.collect()
    print(f"a full collection of this process's heap: {1e3 * (time.perf_counter() - t0):.1f} ms ({n} unreachable)")


if __name__ == "__main__":
    main()
