#!/usr/bin/env python3
"""A/B of the batch-1 decode path in one process: residual + RMSNorm fused into the consuming GEMM vs separate launches."""
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

dev = torch.device("cuda:0")
cfg = ModelConfig()
state = synth.random_state(cfg, seed=0, device=dev)
model = q.Qwen3ASR.from_state_dict(cfg, state, synth.SyntheticTokenizer(), dev)
del state
a1, l1d, l1 = q.audio.pack_audio([signal("speechlike", 16000 * 30, seed=1)], dev)
res = {"qkv": [], "qkv+gu": [], "none": []}
toks = {}
for rnd in range(6):
    for fuse in ("qkv", "qkv+gu", "none"):
        model._decoder.fuse_norm = fuse != "none"
        model._decoder.fuse_norm_gu = fuse == "qkv+gu"
        for ws in model._decoder._ws.values():
            if ws.graph is not None:
                ws.graph_key = None                      # recapture for the other path
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = model.generate_tokens_device(a1, l1d, l1, [128], "English", repetition_penalty=1.2)
        torch.cuda.synchronize()
        if rnd:
            res[fuse].append(time.perf_counter() - t0)
        toks[fuse] = out
for fuse in res:
    print(f"fuse_norm={fuse}: median {1e3 * float(np.median(res[fuse])):.1f} ms  min {1e3 * min(res[fuse]):.1f} ms")
print("same tokens:", toks["qkv"] == toks["none"] == toks["qkv+gu"])
