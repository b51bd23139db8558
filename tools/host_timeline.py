#!/usr/bin/env python3
"""Where the HOST is while one batch-32 request is issued: wall-clock stamps around the phases of generate_tokens_device without
any added synchronisation.  A phase whose host time is close to the GPU time of the work queued before it is a hidden wait.
Round 4: `_merge`'s pageable H2D copy parks the host until the encoder pass has run (merge issued +16.4 ms).  With every small upload
staged in pinned memory the host has issued the prefill by +4.1 ms -- and the prefill ends at +51.0 ms either way (50.8 before): the
GPU was not waiting for the host there.  torch's pinned non-blocking copies also break graph capture on another lane's thread
(hipErrorCapturedEvent from the caching host allocator's event queries), so the pageable copies stay."""
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
import importlib
G = importlib.import_module("qwen3_asr_mlx_amd.generate")
from qwen3_asr_mlx_amd.config import ModelConfig  # noqa: E402
from qwen3_asr_mlx_amd.tokenizer import build_prompt  # noqa: E402
from signals import signal  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0")
cfg = ModelConfig()
state = synth.random_state(cfg, seed=0, device=dev)
model = q.Qwen3ASR.from_state_dict(cfg, state, synth.SyntheticTokenizer(), dev)
del state
audio, lens_dev, lens = q.audio.pack_audio([signal("speechlike", 16000 * 30, seed=1 + i) for i in range(B)], dev)
for _ in range(3):
    model.generate_tokens_device(audio, lens_dev, lens, [128] * B, "English", repetition_penalty=1.2)
torch.cuda.synchronize()

dec, enc = model._decoder, model._encoder
side = G._side_stream(dev)
rows = []
for rep in range(3):
    torch.cuda.synchronize()
    st = [("start", time.perf_counter())]
    mark = lambda name: st.append((name, time.perf_counter()))       # noqa: E731
    with torch.cuda.stream(side):
        fe = q.audio.get_frontend(dev)
        mel, n_frames = fe(audio, lens_dev, lens); mark("mel issued")
        feats, tok_counts = enc.forward_batch(mel, n_frames); mark("encoder issued")
        lang_ids = model._tokenizer.encode(" English")
        prompts = [build_prompt(n, lang_ids) for n in tok_counts]; mark("prompts built")
        seq_lens = [len(p) for p in prompts]
        embeds = G._merge(feats, prompts, tok_counts, dec); mark("merge issued")
        Lmax = (max(t + 128 for t in seq_lens) + 63) // 64 * 64
        cache = dec.batch_cache(B, Lmax)
        ws = dec.workspace(B, 128)
        ws.n_gen.zero_(); ws.finished.zero_()
        ws.pos.copy_(torch.tensor([t - 1 for t in seq_lens], dtype=torch.int32), non_blocking=False)
        ws.max_new.copy_(torch.tensor([128] * B, dtype=torch.int32)); mark("state set")
        sampling = {"temperature": 0.0, "top_p": 1.0, "top_k": 0, "repetition_penalty": 1.2, "repetition_context_size": 100, "seed": 0}
        dec.prefill(embeds, seq_lens, cache, logits_out=ws.logits); mark("prefill issued")
        dec.sample_step(ws, ws.logits, sampling); mark("first token issued")
        side.synchronize(); mark("prefill done (sync)")
    rows.append(st)
for st in rows[1:]:
    t0 = st[0][1]
    print("  ".join(f"{name} +{1e3 * (t - t0):.2f}" for name, t in st[1:]))
