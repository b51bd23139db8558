#!/usr/bin/env python3
"""Where one encoder pass and one prefill spend their time: every C-ABI call bracketed by HIP events on the launch stream
(`_lib.PROFILE`), aggregated by call name and shape.  Usage: python3 tools/stage_profile.py [batch] [seconds] [out.json]"""
import collections
import ctypes as C
import json
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import qwen3_asr_mlx_amd as q  # noqa: E402
from qwen3_asr_mlx_amd import _lib, synth  # noqa: E402
from qwen3_asr_mlx_amd.config import ModelConfig  # noqa: E402
from qwen3_asr_mlx_amd.generate import _merge  # noqa: E402
from qwen3_asr_mlx_amd.tokenizer import build_prompt  # noqa: E402
from signals import signal  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 30.0
dev = torch.device("cuda:0")
cfg = ModelConfig()
model = q.Qwen3ASR.from_state_dict(cfg, synth.random_state(cfg, seed=0, device=dev), synth.SyntheticTokenizer(), dev)
dec, enc = model._decoder, model._encoder
audio, lens_dev, lens = q.audio.pack_audio([signal("speechlike", int(secs * 16000), seed=1234 + i) for i in range(B)], dev)
fe = q.audio.get_frontend(dev)


def shape_of(name, a):
    v = lambda x: x.value if hasattr(x, "value") else x          # noqa: E731
    if name == "q3asr_gemm_bf16":
        return f"M={v(a[4])} N={v(a[5])} K={v(a[6])} act={v(a[8])} res={v(a[10])} out32={v(a[14])}"
    if name == "q3asr_conv3x3s2_bf16":
        return f"imgs={v(a[1])} {v(a[2])}x{v(a[3])}x{v(a[4])} -> Cout={v(a[6])}"
    if name == "q3asr_attn_fwd":
        return f"segs={v(a[12])} max={v(a[13])} hd={v(a[17])} causal={v(a[18])}"
    if name in ("q3asr_layernorm_f32_bf16",):
        return f"M={v(a[4])} D={v(a[5])}"
    if name == "q3asr_rmsnorm_bf16":
        return f"M={v(a[6])} D={v(a[7])}"
    return ""


def profile(fn, label):
    names = set(_lib.EXPORTS)
    fn()                                                        # warm (plans, code objects)
    torch.cuda.synchronize()
    prof = {"names": names, "stream": C.c_void_p(torch.cuda.current_stream().cuda_stream), "events": []}
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    _lib.PROFILE = prof
    e0.record()
    out = fn()
    e1.record()
    _lib.PROFILE = None
    torch.cuda.synchronize()
    agg = collections.OrderedDict()
    for name, ms, args in _lib.drain_profile(prof):
        k = (name.replace("q3asr_", ""), shape_of(name, args))
        a = agg.setdefault(k, [0, 0.0])
        a[0] += 1
        a[1] += ms
    total = sum(v[1] for v in agg.values())
    print(f"--- {label}: {e0.elapsed_time(e1):.2f} ms wall with per-call events, {total:.2f} ms inside calls")
    rows = []
    for (n, s), (cnt, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"  {ms:8.3f} ms  {100 * ms / total:5.1f}%  x{cnt:<4d} {1e3 * ms / cnt:8.1f} us  {n} {s}")
        rows.append({"call": n, "shape": s, "count": cnt, "ms": round(ms, 4), "us_each": round(1e3 * ms / cnt, 2)})
    return out, {"label": label, "ms_in_calls": round(total, 3), "rows": rows}


report = []
(mel, n_frames), r = profile(lambda: fe(audio, lens_dev, lens), "mel")
report.append(r)
(feats, counts), r = profile(lambda: enc.forward_batch(mel, n_frames), f"encoder B={B} {secs:g}s")
report.append(r)
prompts = [build_prompt(c, [6364]) for c in counts]
seq = [len(p) for p in prompts]
cache = dec.batch_cache(B, (max(seq) + 128 + 63) // 64 * 64)


def prefill():
    emb = _merge(feats, prompts, counts, dec)
    return dec.prefill(emb, seq, cache)


_, r = profile(prefill, f"prefill B={B} T={seq[0]}")
report.append(r)
if len(sys.argv) > 3:
    Path(sys.argv[3]).write_text(json.dumps(report, indent=1))
