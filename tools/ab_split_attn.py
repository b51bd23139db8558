#!/usr/bin/env python3
"""A/B of the batch <= 2 decode attention in one process: 64 single-wave workgroups per sequence with the merge in the o
projection's prologue (split) against one 512-thread workgroup per (kv head, sequence) (fused).  Interleaved rounds, same
tokens required.  Usage: ab_split_attn.py [batch (1|2)] [clip seconds] [tokens]"""
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

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 30.0
ntok = int(sys.argv[3]) if len(sys.argv) > 3 else 128
dev = torch.device("cuda:0")
cfg = ModelConfig()
state = synth.random_state(cfg, seed=0, device=dev)
model = q.Qwen3ASR.from_state_dict(cfg, state, synth.SyntheticTokenizer(), dev)
del state
a1, l1d, l1 = q.audio.pack_audio([signal("speechlike", int(16000 * secs), seed=1 + i) for i in range(B)], dev)
res = {"split": [], "fused": []}
toks = {}
for rnd in range(7):
    for arm in ("split", "fused"):
        model._decoder.split_attn = arm == "split"
        for ws in model._decoder._ws.values():
            if ws.graph is not None:
                ws.graph_key = None                      # recapture for the other path
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = model.generate_tokens_device(a1, l1d, l1, [ntok] * B, "English", repetition_penalty=1.2)
        torch.cuda.synchronize()
        if rnd:
            res[arm].append(time.perf_counter() - t0)
        toks[arm] = out
for arm in res:
    print(f"batch {B}, {secs:g} s clips, {ntok} tokens, attention {arm}: median {1e3 * float(np.median(res[arm])):.1f} ms  min {1e3 * min(res[arm]):.1f} ms")
print("same tokens:", toks["split"] == toks["fused"])
