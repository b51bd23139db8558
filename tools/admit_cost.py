#!/usr/bin/env python3
"""What an admission costs the decode rows (serve.ContinuousTranscriber / generate.RowScheduler): mel + encoder and the packed prefill
of k arriving 30 s clips, k = 1, 2, 4, 8, 16 -- wall time on the device against the time the host needs to ISSUE the launches
(the call returning, nothing synchronised).  If the two were close, a captured graph of the group's launches would help; where the
device time is several times the issue time, the group is bound by its small-M kernels, not by launching them."""
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import qwen3_asr_mlx_amd as q  # noqa: E402
from qwen3_asr_mlx_amd import synth  # noqa: E402
from qwen3_asr_mlx_amd.config import ModelConfig  # noqa: E402
from qwen3_asr_mlx_amd.generate import _merge  # noqa: E402
from qwen3_asr_mlx_amd.tokenizer import build_prompt  # noqa: E402
from signals import signal  # noqa: E402

dev = torch.device("cuda:0")
cfg = ModelConfig()
model = q.Qwen3ASR.from_state_dict(cfg, synth.random_state(cfg, seed=0, device=dev), synth.SyntheticTokenizer(), dev)
dec, enc = model._decoder, model._encoder
fe = q.audio.get_frontend(dev)
cache = dec.batch_cache(32, 576)
for k in (1, 2, 4, 8, 16):
    clips = [signal("speechlike", 480000, seed=77 + i) for i in range(k)]
    audio, lens_dev, lens = q.audio.pack_audio(clips, dev)
    rows = list(range(k))

    def encode():
        mel, nf = fe(audio, lens_dev, lens)
        return enc.forward_batch(mel, nf)

    feats, counts = encode()
    prompts = [build_prompt(c, [6364]) for c in counts]
    seq = [len(p) for p in prompts]
    logits = torch.empty((k, cfg.text_decoder.vocab_size), dtype=torch.bfloat16, device=dev)

    def prefill():
        emb = _merge(feats, prompts, counts, dec)
        dec.prefill(emb, seq, cache, logits_out=logits, slots=rows)

    out = {}
    for name, fn in (("encoder", encode), ("prefill", prefill)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        issue, wall = [], []
        for _ in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            issue.append(t1 - t0)
            wall.append(t2 - t0)
        out[name] = (1e3 * sorted(issue)[2], 1e3 * sorted(wall)[2])
    print(f"k={k:2d}: encoder issue {out['encoder'][0]:.2f} ms / device {out['encoder'][1]:.2f} ms; prefill issue {out['prefill'][0]:.2f} ms / "
          f"device {out['prefill'][1]:.2f} ms  ({out['prefill'][1] / k:.2f} ms per sequence)", flush=True)
