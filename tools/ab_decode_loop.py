#!/usr/bin/env python3
"""What the decode loop's host side costs a batch-32 request: steps per graph replay (MULTI_STEPS) and the cadence of the
`finished` check (a device-to-host synchronisation), A/B in one process, interleaved rounds, same tokens."""
import importlib
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
from signals import signal  # noqa: E402
gen = importlib.import_module("qwen3_asr_mlx_amd.generate")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0")
cfg = ModelConfig()
state = synth.random_state(cfg, seed=0, device=dev)
model = q.Qwen3ASR.from_state_dict(cfg, state, synth.SyntheticTokenizer(), dev)
del state
a, ld, l = q.audio.pack_audio([signal("speechlike", 16000 * 30, seed=i) for i in range(B)], dev)
arms = [tuple(a_.split(":")) for a_ in (sys.argv[2] if len(sys.argv) > 2 else "4:16,8:16,16:16,4:1000,16:1000").split(",")]
arms = [(int(a_[0]), int(a_[1]), a_[2] if len(a_) > 2 else "look", len(a_) > 3 and a_[3] == "pp") for a_ in arms]
# fourth field "pp": two instantiations of the long graph, replayed alternately (is a replay of an executable graph that is still
# in flight what costs?)
# what a "look" does (third field): look = reduction + synchronising read (the product); sync = stream synchronise only;
# kernel = the reduction kernel without waiting for it; query = hipStreamQuery (non-blocking); none = nothing
looks = {"look": gen._all_finished,
         "sync": lambda ws: (torch.cuda.current_stream().synchronize(), False)[1],
         "kernel": lambda ws: (ws.finished.all(), False)[1],
         "query": lambda ws: (torch.cuda.current_stream().query(), False)[1],      # non-blocking: lets the runtime retire what has finished
         "none": lambda ws: False}
res, toks = {k: [] for k in arms}, {}
dec = model._decoder
from qwen3_asr_mlx_amd import _lib  # noqa: E402
import ctypes as C  # noqa: E402
PP = {"on": False}
_capture_step, _replay_step = dec.capture_step, dec.replay_step


def capture_step(ws, cache, sampling):
    ws._pp_args = (cache, sampling)
    return _capture_step(ws, cache, sampling)


def replay_step(ws, multi=False):
    if PP["on"] and multi and ws.graph_multi is not None:
        if getattr(ws, "_pp_graph", None) is None:
            ws._pp_graph = dec._capture(ws, ws._pp_args[0], ws._pp_args[1], dec.MULTI_STEPS)
            ws._pp_flip = 0
        ws._pp_flip ^= 1
        g = ws._pp_graph if ws._pp_flip else ws.graph_multi
        _lib.call("q3asr_graph_launch", g, C.c_void_p(torch.cuda.current_stream().cuda_stream))
        return dec.MULTI_STEPS
    return _replay_step(ws, multi=multi)


dec.capture_step, dec.replay_step = capture_step, replay_step
for rnd in range(7):
    for arm in arms:
        dec.MULTI_STEPS, gen.FINISH_CHECK_EVERY = arm[0], arm[1]
        gen._all_finished = looks[arm[2]]
        PP["on"] = arm[3]
        for ws in dec._ws.values():
            if getattr(ws, "_pp_graph", None) is not None:
                _lib.call("q3asr_graph_destroy", ws._pp_graph)
                ws._pp_graph = None
            if ws.graph_multi is not None:
                _lib.call("q3asr_graph_destroy", ws.graph_multi)
                ws.graph_multi = None                    # re-captured with the arm's step count on the next request
        for _ in range(2):                               # the second request of an arm runs on the long graph
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = model.generate_tokens_device(a, ld, l, [128] * B, "English", repetition_penalty=1.2)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        if rnd:
            res[arm].append(dt)
        toks[arm] = out
for arm in arms:
    print(f"steps per replay {arm[0]:2d}, {arm[2]:6s} every {arm[1]:4d} steps{' (two graphs alternating)' if arm[3] else ''}: median {1e3 * float(np.median(res[arm])):.2f} ms  min {1e3 * min(res[arm]):.2f} ms")
print("same tokens:", all(toks[a_] == toks[arms[0]] for a_ in arms))
